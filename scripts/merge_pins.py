#!/usr/bin/env python3
"""Merge outputs of scripts/pin_fullsize_ref.py (gpurun_out/<tag>/pin_*.json) into tests/golden/bench_checksums_ref.json.

    python3 scripts/merge_pins.py gpurun_out/r6a/pin_*.json

Only runs whose arrays were equal element for element are taken; the key is 'workload/k/w' for a whole workload and
'workload/k/w@G' for its first G genomes.  Everything in the file comes from the compiled reference's arrays.
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DST = ROOT / "tests" / "golden" / "bench_checksums_ref.json"
SCHEME = 2   # r06: every field of a node / edge mixed with its element index, penalty by bit pattern (csrc/device.hpp ck_*)


def main():
    cur = json.loads(DST.read_text()) if DST.exists() else {}
    drop_stale = "--drop-stale" in sys.argv
    for p in [a for a in sys.argv[1:] if not a.startswith("--")]:
        e = json.loads(Path(p).read_text())
        if not (e.get("equal") and all(e["hip_vs_reference_elementwise"].values())):
            print("SKIP (not equal):", p)
            continue
        if e.get("checksum_scheme") != SCHEME:
            print("SKIP (checksum scheme):", p)
            continue
        key = f"{e['workload']}/k{e['k']}/w{e['w']}" + ("" if e["genomes"] == e["genomes_of_workload"] else f"@{e['genomes']}")
        cur[key] = e
        print("merged", key, e["checksums"], e["counts"])
    stale = [k for k, v in cur.items() if k != "_about" and v.get("checksum_scheme") != SCHEME]
    for k in stale:
        print("stale entry (old checksum scheme):", k, "-- dropped" if drop_stale else "-- kept (tests refuse it; --drop-stale removes it)")
        if drop_stale:
            del cur[k]
    DST.write_text(json.dumps(cur, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
