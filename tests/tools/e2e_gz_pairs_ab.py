#!/usr/bin/env python3
"""Level-6 .fa.gz -> numpy with the workers inflating two files in one loop (r06, SEQWIN_AMD_GZ_PAIRS=1) and one file at a time (the
default), alternating, on genomes of the default workload: wall time and split per call, arrays compared.

    python3 tests/tools/e2e_gz_pairs_ab.py [genomes] [n_cpu] [repeats]"""
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np

    from bench import SEED, WORKLOADS, _gzip_one, e2e_build, make_batch, write_fasta_fast
    from seqwin_amd.device import set_device
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    set_device(0)
    b = make_batch(WORKLOADS["bacteria15k"], G, SEED)
    tmp = tempfile.mkdtemp(prefix="seqwin_gzab_", dir="/dev/shm")
    try:
        paths, bp = write_fasta_fast(b, G, tmp, min(32, os.cpu_count() or 1))
        b.close()
        gz = [os.path.join(tmp, f"z{a}.fa.gz") for a in range(G)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(min(64, os.cpu_count() or 1)) as pool:
            sizes = list(pool.map(_gzip_one, list(zip(paths, gz))))
        for p in paths:
            os.unlink(p)
        print(f"{G} files, {bp / 1e9:.2f} Gbp, gzipped (level 6) to {sum(sizes) / 1e9:.2f} GB in {time.perf_counter() - t0:.0f} s; n_cpu {n_cpu}", flush=True)
        tar = np.arange(G) % 2 == 0
        e2e_build(gz[:8], 21, 200, 4, tar[:8])
        ref = None
        for rep in range(reps):
            for name, env in (("pairs", "1"), ("one file", None)):
                os.environ.pop("SEQWIN_AMD_GZ_PAIRS", None)
                if env:
                    os.environ["SEQWIN_AMD_GZ_PAIRS"] = env
                got, wall, split = e2e_build(gz, 21, 200, n_cpu, tar)
                if ref is None:
                    ref, eq = got, True
                else:
                    eq = all(np.array_equal(a, c) for a, c in zip(got, ref))
                    del got
                print(f"rep {rep} {name:9s} {bp / wall / 1e9:6.2f} Gbp/s  wall {wall * 1e3:7.1f} ms  ingest+upload {split['ingest_upload_ms']:7.1f}  "
                      f"device(exposed) {split['device_ms']:6.1f}  export {split['export_ms']:5.1f}  equal {eq}", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
