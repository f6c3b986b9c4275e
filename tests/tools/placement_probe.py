#!/usr/bin/env python3
"""Is the spread of the nodes stage (46.8 ... 50.1 ms at 15 000 genomes, constant inside a process, different between processes of
one library on one box: NOTES.md) a property of the PROCESS or of where its blocks physically lie?  The virtual layout of the pool's
blocks is the same in every process (scripts/gpu/bimodal.sh: every block at the same offset from the first, only the base moves), so
the test is: inside ONE process, build a few times, hand every cached block back to the driver (sw_pool_trim: the next build's
hipMallocs get new physical pages), build again -- several rounds.  If the stage time moves between rounds and not inside a round,
it is the physical placement the driver chooses, which no user-space skew of the buffers can steer.

    python3 tests/tools/placement_probe.py [rounds] [builds per round]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np

    from bench import SEED, WORKLOADS, make_batch
    from seqwin_amd.device import pool_trim, set_device
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    set_device(0)
    G = WORKLOADS["bacteria15k"][0]
    b = make_batch(WORKLOADS["bacteria15k"], G, SEED)
    tar = np.arange(G) % 2 == 0
    for r in range(rounds):
        row = []
        for i in range(per + 1):
            ix = b.build_index(21, 200, tar)
            t = ix.timings()
            ix.close()
            if i:   # (the first build of a round pays the hipMallocs)
                row.append((t["sketch_ms"], t["nodes_ms"], t["edges_ms"], t["total_ms"]))
        print(f"round {r}: nodes " + " ".join(f"{x[1]:.2f}" for x in row) + "   edges " + " ".join(f"{x[2]:.2f}" for x in row) + "   sketch "
              + " ".join(f"{x[0]:.2f}" for x in row) + "   total " + " ".join(f"{x[3]:.2f}" for x in row), flush=True)
        pool_trim()


if __name__ == "__main__":
    main()
