#!/usr/bin/env python3
"""What ONE sw_build per process costs at full size -- the situation of the `seqwin` CLI: a fresh interpreter, FASTA -> numpy once.
The FASTA files of the workload are written once; then every variant (environment) runs in its own process, twice in a row.

    python3 tests/tools/e2e_first_call.py [genomes] [n_cpu]"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from bench import e2e_build
paths = [l.strip() for l in open(sys.argv[2])]
n_cpu = int(sys.argv[3])
tar = np.arange(len(paths)) % 2 == 0
warm_s = None
if os.environ.get("FIRST_CALL_PREWARM") == "1":   # a tiny build first: every code object of the library gets loaded
    from seqwin_amd.device import Batch
    tw = time.perf_counter()
    b = Batch.synthetic(4, 2, 3000, n_ancestors=2, snp_ppm=10000, seed=1)
    b.build_index(21, 200, [True, False, True, False]).close()
    b.close()
    warm_s = round(time.perf_counter() - tw, 3)
t0 = time.perf_counter()
got, wall, split = e2e_build(paths, 21, 200, n_cpu, tar)
print(json.dumps({"prewarm_s": warm_s, "wall_s": round(wall, 3), "since_start_s": round(time.perf_counter() - t0, 3), "split": split, "kmers": int(len(got[0]))}))
"""


def main():
    from bench import SEED, WORKLOADS, make_batch, write_fasta_fast
    from seqwin_amd.device import set_device
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
    n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    set_device(0)
    b = make_batch(WORKLOADS["bacteria15k"], G, SEED)
    tmp = tempfile.mkdtemp(prefix="seqwin_first_", dir="/dev/shm")
    try:
        paths, bp = write_fasta_fast(b, G, tmp, min(64, os.cpu_count() or 1))
        b.close()
        lst = os.path.join(tmp, "paths.txt")
        open(lst, "w").write("\n".join(paths) + "\n")
        variants = [("default", {}), ("default again", {}), ("pinned pool 128 MB", {"SEQWIN_AMD_PINNED_POOL_MB": "128"}),
                    ("pinned pool 0 (ring only)", {"SEQWIN_AMD_PINNED_POOL_MB": "0"}), ("default, third", {})]
        if os.environ.get("FIRST_CALL_VARIANTS"):   # "name=ENV=VAL,ENV2=VAL2;name2=..."
            variants = []
            for item in os.environ["FIRST_CALL_VARIANTS"].split(";"):
                name, _, envs = item.partition("=")
                variants.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
        import time
        for name, env in variants:
            time.sleep(float(os.environ.get("FIRST_CALL_PAUSE", "0")))   # (let whatever the previous process left behind settle)
            e = dict(os.environ, SEQWIN_AMD_DEBUG_TIMING="1", **env)
            r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), lst, str(n_cpu)], capture_output=True, text=True, env=e, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print(name, "FAILED", r.stderr[-800:])
                continue
            d = json.loads(line[-1])
            s = d["split"]
            print(f"{name:28s} prewarm {d.get('prewarm_s')} wall {d['wall_s']:6.2f} s = {bp / d['wall_s'] / 1e9:5.1f} Gbp/s  ingest+upload {s['ingest_upload_ms']:7.0f} ms  device {s['device_ms']:6.0f}  export {s['export_ms']:6.0f}  "
                  f"penalty {s['get_penalty_wall_ms']:5.0f}  cpu_s {s['cpu_s']:6.1f}  throttled {s['quota_throttled_ms']:8.0f} ms | on the device: plan {s['plan_ms']:.0f} sketch {s['sketch_ms']:.0f} "
                  f"nodes {s['nodes_ms']:.0f} edges {s['edges_ms']:.0f}", flush=True)
            dbg = [ln for ln in r.stderr.splitlines() if "seqwin_amd" in ln]
            for ln in dbg[:12]:
                print("      ", ln[:200])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
