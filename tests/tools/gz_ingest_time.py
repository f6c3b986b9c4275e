"""Times FASTA.gz -> device batch through the host route (zlib on n_cpu threads + SIMD packer) and through the device route
(csrc/ingest_dev.hip) on synthetic genomes.  usage: python tests/tools/gz_ingest_time.py [n_files] [mbp_per_file] [n_cpu]"""
import gzip
import os
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))


def make(args):
    path, seed, n_bp = args
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_bp)]
    lines = []
    per = n_bp // 4                      # four contigs of 80-column lines
    for c in range(4):
        body = seq[c * per:(c + 1) * per]
        pad = (-len(body)) % 80
        rows = np.concatenate([body, np.full(pad, ord("A"), np.uint8)]).reshape(-1, 80)
        rows = np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1)
        lines.append(f">contig{c} len={per}\n".encode() + rows.tobytes())
    with gzip.GzipFile(path, "wb", compresslevel=6, mtime=0) as f:
        f.write(b"".join(lines))
    return os.path.getsize(path)


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    n_cpu = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    d = tempfile.mkdtemp(prefix="gzt_", dir=os.environ.get("TMPDIR", "/tmp"))
    paths = [os.path.join(d, f"g{i}.fa.gz") for i in range(n_files)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=n_cpu) as ex:
        comp = sum(ex.map(make, [(p, i, int(mbp * 1e6)) for i, p in enumerate(paths)], chunksize=8))
    print(f"{n_files} files x {mbp} Mbp: {comp / 1e6:.0f} MB compressed, written in {time.perf_counter() - t0:.1f} s", flush=True)
    from seqwin_amd.device import Batch
    res = {}
    for rep in range(2):
        for route in ("0", "1"):
            os.environ["SEQWIN_AMD_DEVICE_INFLATE"] = route
            t0 = time.perf_counter()
            b = Batch.from_fasta(paths, n_cpu=n_cpu)
            dt = time.perf_counter() - t0
            info = b.info()
            res[route] = (dt, info["total_bp"], info["n_records"])
            print(f"rep {rep} route {'device' if route == '1' else 'host'}: {dt * 1e3:.0f} ms, {info['total_bp'] / dt / 1e9:.2f} Gbp/s", flush=True)
            if rep == 1:
                ck = b.build_index(21, 200).checksums()
                res[route] += (ck,)
            del b
    assert res["0"][1:] == res["1"][1:], (res["0"], res["1"])
    print("same batch (index checksums equal)")
    for p in paths:
        os.remove(p)
    os.rmdir(d)


if __name__ == "__main__":
    main()
