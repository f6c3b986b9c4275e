import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from seqwin_amd import dist as swdist
eng = swdist.HipEngine()
rng = np.random.default_rng(1)
for bits in (32, 8, 3, 1):
    os.environ["SEQWIN_AMD_SORT_KEYBITS"] = str(bits)
    for n in (50, 1000, 5000, 70000, 300000):
        for kind in ("random", "dups"):
            if kind == "random":
                h = rng.integers(0, 2**64, n, dtype=np.uint64)
            else:
                vals = rng.integers(0, 2**64, max(2, n // 20), dtype=np.uint64)
                h = vals[rng.integers(0, len(vals), n)]
            kmer = np.arange(n, dtype=np.uint64)
            rows = torch.from_numpy(np.stack([h, kmer], axis=1).view(np.int64))
            ix, ranks = eng.slice_build(rows.to(eng.gpu), 0, np.array([0, 1], np.uint32), None)
            K, N, E = ix.export()
            order = np.argsort(h, kind="stable")
            ok = np.array_equal(K["pos"], order.astype(np.uint32))
            hs_ok = np.array_equal(h[K["pos"]], h[order])
            print(bits, n, kind, "OK" if ok else "BAD", "hashes sorted" if hs_ok else "hashes NOT sorted", flush=True)
            if not ok:
                d = np.nonzero(K["pos"] != order.astype(np.uint32))[0]
                print("   ", len(d), "diffs, first", d[:5], K["pos"][d[:5]], order[d[:5]], [hex(x) for x in h[order[d[:5]]]])
