#!/usr/bin/env python3
"""The multi-device form INSIDE one sw_build (SEQWIN_DEVICES, csrc/multi.hip) at 1 / 2 / 4 / 8 devices: FASTA files in /dev/shm ->
sw_build + sw_graph_export, arrays compared with the single-device build's, wall time and the library's own split per N.

    python3 tests/tools/multi_device_scale.py OUTDIR [MAX_DEVICES] [GENOMES]

One JSON line per N in OUTDIR/devices_n<N>.json.  SCALE_LOGICAL=1 lists device 0 N times (a rehearsal on one card: the lines
then say "logical").  Needs a GPU; part of scripts/scale.sh."""
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np

    from bench import SEED, WORKLOADS, e2e_build, write_fasta_fast
    from seqwin_amd.device import Batch, device_count, set_device
    out = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/scale")
    max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    out.mkdir(parents=True, exist_ok=True)
    logical = os.environ.get("SCALE_LOGICAL") == "1"
    n_dev = device_count()
    _, rpg, rl, anc, snp, _ = WORKLOADS["bacteria15k"]
    set_device(0)
    b = Batch.synthetic(G, rpg, rl, n_ancestors=anc, snp_ppm=snp, seed=SEED)
    tmp = tempfile.mkdtemp(prefix="seqwin_scale_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        paths, bp = write_fasta_fast(b, G, tmp, min(32, os.cpu_count() or 1))
        b.close()
        tar = np.arange(G) % 2 == 0
        n_cpu = min(64, os.cpu_count() or 1)
        ref = None
        for n in (1, 2, 4, 8):
            if n > max_n or (n > n_dev and not logical):
                continue
            if n == 1:
                os.environ.pop("SEQWIN_DEVICES", None)
            else:
                os.environ["SEQWIN_DEVICES"] = ",".join("0" if logical else str(d) for d in range(n))
            e2e_build(paths[:4], 21, 200, 4, tar[:4])                     # warm-up: streams, pools, rings of every device
            best = None
            for _ in range(3):
                got, wall, split = e2e_build(paths, 21, 200, n_cpu, tar)
                if best is None or wall < best[1]:
                    best = (got, wall, split)
                else:
                    del got
            got, wall, split = best
            if ref is None:
                ref = got
                equal = True
            else:
                equal = all(np.array_equal(a, c) for a, c in zip(got, ref))
            line = {"form": "inside one sw_build (SEQWIN_DEVICES)", "n_devices": n, "logical": logical, "genomes": G, "Gbp": round(bp / 1e9, 2),
                    "wall_s": round(wall, 4), "Gbp_per_s_fasta_to_numpy": round(bp / wall / 1e9, 2), "split_ms": split,
                    "arrays_equal_to_single_device": bool(equal), "n_cpu": n_cpu}
            (out / f"devices_n{n}.json").write_text(json.dumps(line) + "\n")
            print(json.dumps(line))
            if got is not ref:
                del got
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
