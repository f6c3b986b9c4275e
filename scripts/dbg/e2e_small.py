import os, sys, time, glob
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from seqwin_amd import _core
from seqwin_amd.device import Batch
n_genomes = 256
b = Batch.synthetic(n_genomes, 50, 96000, n_ancestors=5, snp_ppm=10000, seed=20260821)
offs, ids = b.records()
tmp = "/dev/shm/e2e_x"; os.makedirs(tmp, exist_ok=True)
paths = []
for a in range(n_genomes):
    p = os.path.join(tmp, f"g{a}.fa")
    with open(p, "wb") as f:
        for r in range(int(offs[a]), int(offs[a + 1])):
            s = b.record(r)
            f.write(b">" + ids[a][r - int(offs[a])].encode() + b"\n")
            f.write(b"\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + b"\n")
    paths.append(p)
for n_cpu in (4, 8, 16):
    for rep in range(2):
        t0 = time.perf_counter(); bb = Batch.from_fasta(paths, n_cpu=n_cpu); t1 = time.perf_counter()
        print(f"from_fasta n_cpu={n_cpu}: {(t1-t0)*1e3:.1f} ms", flush=True)
        bb.close()
    os.environ["SEQWIN_AMD_NO_STREAM_UPLOAD"] = "1"
    t0 = time.perf_counter(); bb = Batch.from_fasta(paths, n_cpu=n_cpu); t1 = time.perf_counter()
    print(f"from_fasta n_cpu={n_cpu} (no stream): {(t1-t0)*1e3:.1f} ms", flush=True)
    bb.close()
    del os.environ["SEQWIN_AMD_NO_STREAM_UPLOAD"]
for p in paths: os.unlink(p)
