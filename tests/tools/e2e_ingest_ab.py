"""A/B of host-ingest variants on the GPU box: the first G genomes of the default workload as plain FASTA in /dev/shm through
sw_build + sw_graph_export + sw_get_penalty (bench.py's e2e leg), per environment setting.  usage: e2e_ingest_ab.py [G] [n_cpu]"""
import os, sys, tempfile, time, shutil
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from bench import SEED, WORKLOADS, e2e_build, write_fasta_fast
from seqwin_amd.device import Batch, set_device

if os.environ.get("AB_TORCH"):        # as bench.py: torch (and its threads) in the process
    import torch
    torch.cuda.init()
    torch.zeros(1, device="cuda")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 64
_, rpg, rl, anc, snp, _ = WORKLOADS["bacteria15k"]
set_device(0)
b = Batch.synthetic(G, rpg, rl, n_ancestors=anc, snp_ppm=snp, seed=SEED)
tmp = tempfile.mkdtemp(prefix="e2e_ab_", dir="/dev/shm")
paths, bp = write_fasta_fast(b, G, tmp, 32)
b.close()
tar = np.arange(G) % 2 == 0
if os.environ.get("AB_REF"):          # as bench.py: the compiled reference has run on the files first (all hardware threads, then 8)
    import oracle
    ref = oracle.load_ref()
    for nc in (os.cpu_count(), 8):
        t0 = time.perf_counter()
        ref._build_native(paths[:512], 21, 200, nc, False)
        print(f"reference on 512 files at {nc} threads: {time.perf_counter() - t0:.1f} s", flush=True)
e2e_build(paths[:2], 21, 200, 2, tar[:2])
try:
    for rep in range(int(os.environ.get("AB_REPS", "2"))):
        modes = (("mmap", {"SEQWIN_AMD_MMAP": "1"}), ("read()", {"SEQWIN_AMD_MMAP": "0"}), ("mmap+populate", {"SEQWIN_AMD_MMAP": "2"}))
        if os.environ.get("AB_MODES") == "pinned":   # r05: the parsers' buffers page-locked (default) / the ring for every copy / no window either
            modes = (("pinned+window", {}), ("ring+window", {"SEQWIN_AMD_PINNED_POOL_MB": "0"}),
                     ("ring, no window", {"SEQWIN_AMD_PINNED_POOL_MB": "0", "SEQWIN_AMD_INGEST_WINDOW": "1000000"}))
        for name, env in modes:
            for k_, v in env.items():
                os.environ[k_] = v
            _, wall, split = e2e_build(paths, 21, 200, n_cpu, tar)
            for k_ in env:
                os.environ.pop(k_)
            print(f"{name:14s} n_cpu={n_cpu}: {bp / wall / 1e9:6.2f} Gbp/s  wall {wall * 1e3:6.1f} ms  ingest+upload {split['ingest_upload_ms']:6.1f} ms", flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
