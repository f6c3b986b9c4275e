"""T_e2e: FASTA files on local disk -> numpy arrays through the drop-in boundary (sw_build), next to the
reference CPU path on the same files.  Not the bench metric (which starts with inputs resident in HBM)."""
import os, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle
from seqwin_amd import _core
from seqwin_amd.device import Batch

n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = Batch.synthetic(n_genomes, 50, 96000, n_ancestors=5, snp_ppm=10000, seed=20260821)
offs, ids = b.records()
tmp = tempfile.mkdtemp(prefix="e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
paths, bp = [], 0
for a in range(n_genomes):
    p = os.path.join(tmp, f"g{a}.fa")
    with open(p, "wb") as f:
        for r in range(int(offs[a]), int(offs[a + 1])):
            s = b.record(r); bp += len(s)
            f.write(b">" + ids[a][r - int(offs[a])].encode() + b"\n")
            for i in range(0, len(s), 80):
                f.write(s[i:i + 80] + b"\n")
    paths.append(p)
tar = np.arange(n_genomes) % 2 == 0
cores = os.cpu_count()
for n_cpu in (4, 8, 16, 32, 64, cores):
    _core._build_native(paths[:4], 21, 200, n_cpu, False)  # warm
    t0 = time.perf_counter()
    k, n, e, o, _ = _core._build_native(paths, 21, 200, n_cpu, False)
    t1 = time.perf_counter()
    _core._get_penalty_native(k, n, o, tar, n_cpu)
    dt = time.perf_counter() - t0
    print(f"HIP  e2e n_cpu={n_cpu:3d}: {bp/dt/1e9:7.3f} Gbp/s ({dt:.3f} s = build {t1-t0:.3f} + get_penalty {dt-(t1-t0):.3f}; "
          f"{bp/1e6:.0f} Mbp, {len(k)} kmers)")
ref = oracle.load_ref()
if ref is not None:
    for n_cpu in (8, cores):
        t0 = time.perf_counter()
        k2, n2, e2, o2, _ = ref._build_native(paths, 21, 200, n_cpu, False)
        ref._get_penalty_native(k2, n2, o2, tar, n_cpu)
        dt = time.perf_counter() - t0
        print(f"REF  e2e n_cpu={n_cpu:3d}: {bp/dt/1e9:7.3f} Gbp/s ({dt:.3f} s)")
    assert np.array_equal(k, k2) and np.array_equal(n, n2) and np.array_equal(e, e2)
    print("HIP == reference on these files (kmers, scored nodes, edges)")
for p in paths: os.unlink(p)
os.rmdir(tmp)
