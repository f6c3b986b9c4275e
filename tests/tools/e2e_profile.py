import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from seqwin_amd import _core
from seqwin_amd.device import Batch
n_genomes = 256
b = Batch.synthetic(n_genomes, 50, 96000, n_ancestors=5, snp_ppm=10000, seed=20260821)
offs, ids = b.records()
tmp = "/dev/shm/e2e_p"; os.makedirs(tmp, exist_ok=True)
paths = []
for a in range(n_genomes):
    p = os.path.join(tmp, f"g{a}.fa")
    with open(p, "wb") as f:
        for r in range(int(offs[a]), int(offs[a + 1])):
            s = b.record(r)
            f.write(b">" + ids[a][r - int(offs[a])].encode() + b"\n" + b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + b"\n")
    paths.append(p)
b.close()
_core._build_native(paths, 21, 200, 32, False)
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    out = _core._build_native(paths, 21, 200, 32, False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
for p in paths: os.unlink(p)
