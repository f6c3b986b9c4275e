#!/usr/bin/env python3
"""Level-6 .fa.gz -> numpy (host route) with TWO builds of the library on the same files of the same box: an A/B of the host gzip
decoder (or of anything else on the .gz path).  The files are made once; every library runs in a process of its own
(SEQWIN_AMD_LIB), alternating, and prints wall time, split and a checksum of the arrays.

    python3 tests/tools/e2e_gz_lib_ab.py <genomes> <n_cpu> <repeats> <libA.so> <libB.so>
    python3 tests/tools/e2e_gz_lib_ab.py --run <dir> <n_cpu> <repeats>        (one library on the files of <dir>: used by the above)"""
import os
import shutil
import subprocess
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def run(directory, n_cpu, reps):
    import numpy as np

    from bench import e2e_build
    from seqwin_amd._lib import lib
    from seqwin_amd.device import set_device
    set_device(0)
    gz = sorted((str(p) for p in Path(directory).glob("z*.fa.gz")), key=lambda s: int(Path(s).name[1:].split(".")[0]))
    bp = int(open(os.path.join(directory, "bp.txt")).read())
    tar = np.arange(len(gz)) % 2 == 0
    os.environ["SEQWIN_AMD_DEVICE_INFLATE"] = "0"
    e2e_build(gz[:8], 21, 200, 4, tar[:8])
    e2e_build(gz, 21, 200, n_cpu, tar)      # (the process's first full call: pools, page-locked buffers)
    for rep in range(reps):
        got, wall, split = e2e_build(gz, 21, 200, n_cpu, tar)
        crc = 0
        for a in got[:3]:
            crc = zlib.crc32(a.view(np.uint8).reshape(-1)[:1 << 26], crc)
        print(f"  {lib.sw_version().decode()[:40]:40s} rep {rep}: {bp / wall / 1e9:6.2f} Gbp/s  wall {wall * 1e3:7.1f} ms  ingest+upload {split['ingest_upload_ms']:7.1f}  "
              f"device {split['device_ms']:6.1f}  export {split['export_ms']:5.1f}  ids {split.get('ids_list_ms', 0):5.1f}  penalty {split['get_penalty_wall_ms']:5.1f}  "
              f"cpu_s {split['cpu_s']:5.2f}  arrays crc {crc:08x} ({len(got[0])} kmers)", flush=True)
        del got


def main():
    if sys.argv[1] == "--run":
        return run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    G, n_cpu, reps, libs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:]
    from bench import SEED, WORKLOADS, _gzip_one, make_batch, write_fasta_fast
    from seqwin_amd.device import set_device
    set_device(0)
    tmp = tempfile.mkdtemp(prefix="seqwin_gzab_", dir="/dev/shm")
    try:
        b = make_batch(WORKLOADS["bacteria15k"], G, SEED)
        paths, bp = write_fasta_fast(b, G, tmp, min(32, os.cpu_count() or 1))
        b.close()
        gz = [os.path.join(tmp, f"z{a}.fa.gz") for a in range(G)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(min(64, os.cpu_count() or 1)) as pool:
            sizes = list(pool.map(_gzip_one, list(zip(paths, gz))))
        for p in paths:
            os.unlink(p)
        open(os.path.join(tmp, "bp.txt"), "w").write(str(bp))
        print(f"{G} files, {bp / 1e9:.2f} Gbp, gzipped (level 6) to {sum(sizes) / 1e9:.2f} GB in {time.perf_counter() - t0:.0f} s; n_cpu {n_cpu}", flush=True)
        for round_ in range(2):
            for path in libs:
                print(f"{path} (round {round_}):", flush=True)
                env = dict(os.environ, SEQWIN_AMD_LIB=str(Path(path).resolve()))
                subprocess.run([sys.executable, __file__, "--run", tmp, str(n_cpu), str(reps)], env=env, check=False)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
