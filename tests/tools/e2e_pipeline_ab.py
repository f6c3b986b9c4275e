#!/usr/bin/env python3
"""FASTA -> numpy with and without SEQWIN_AMD_PIPELINE (ingest and sketch overlapped inside sw_build), alternating, on the e2e sample of
bench.py (2 048 genomes of the default workload = 10.24 Gbp): wall time and split per call.

    python3 tests/tools/e2e_pipeline_ab.py [genomes] [n_cpu] [repeats]"""
import json
import os
import shutil
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np

    from bench import SEED, WORKLOADS, e2e_build, make_batch, write_fasta_fast
    from seqwin_amd.device import set_device
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    set_device(0)
    b = make_batch(WORKLOADS["bacteria15k"], G, SEED)
    tmp = tempfile.mkdtemp(prefix="seqwin_pipe_", dir="/dev/shm")
    try:
        paths, bp = write_fasta_fast(b, G, tmp, min(32, os.cpu_count() or 1))
        b.close()
        tar = np.arange(G) % 2 == 0
        e2e_build(paths[:8], 21, 200, 4, tar[:8])
        ref = None
        modes = [("off", {}), ("on/8 chunks", {"SEQWIN_AMD_PIPELINE": "1"}), ("on/1024 Mbp", {"SEQWIN_AMD_PIPELINE": "1", "SEQWIN_AMD_PIPELINE_CHUNK_MBP": "1024"}),
                 ("on/4 chunks", {"SEQWIN_AMD_PIPELINE": "1", "SEQWIN_AMD_PIPELINE_CHUNK_MBP": str(int(bp * 81 / 80 / 4) >> 20)})]
        for rep in range(reps):
            for name, env in modes:
                for k_ in ("SEQWIN_AMD_PIPELINE", "SEQWIN_AMD_PIPELINE_CHUNK_MBP"):
                    os.environ.pop(k_, None)
                os.environ.update(env)
                got, wall, split = e2e_build(paths, 21, 200, n_cpu, tar)
                if ref is None:
                    ref = got
                    eq = True
                else:
                    eq = all(np.array_equal(a, c) for a, c in zip(got, ref))
                    del got
                print(f"rep {rep} {name:12s} {bp / wall / 1e9:6.2f} Gbp/s  wall {wall * 1e3:7.1f} ms  ingest+upload {split['ingest_upload_ms']:7.1f}  device(exposed) {split['device_ms']:6.1f}  "
                      f"sketch {split['sketch_ms']:5.1f} nodes {split['nodes_ms']:5.1f} edges {split['edges_ms']:5.1f}  export {split['export_ms']:5.1f}  penalty {split['get_penalty_wall_ms']:5.1f}  equal {eq}",
                      flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
