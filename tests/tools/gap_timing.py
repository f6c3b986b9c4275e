# timing of the sketch on genomes with invalid-base gaps (scaffolds): 64 genomes x 1 record x 4.8 Mbp, 20 N-runs each
import sys, time, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from seqwin_amd.device import Batch
import tempfile
d = tempfile.mkdtemp()
rng = np.random.default_rng(5)
paths = []
for g in range(64):
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), 4_800_000)
    for s in rng.integers(0, 4_790_000, 20):
        seq[s:s + int(rng.integers(1, 2000))] = ord("N")
    p = os.path.join(d, f"g{g}.fa")
    with open(p, "wb") as f:
        f.write(b">s\n" + seq.tobytes() + b"\n")
    paths.append(p)
b = Batch.from_fasta(paths)
for it in range(3):
    ix = b.build_index(21, 200, np.arange(64) % 2 == 0)
    t = ix.timings()
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items() if k in ("sketch_ms", "total_ms", "n_tiles", "ovf_tiles", "total_bp")}, ix.sizes())
    ix.close()
