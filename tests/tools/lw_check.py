"""Route for windows above the split (SEQWIN_AMD_WINDOW_SPLIT, default SW_MAX_WINDOW): sizes and time on a configs[1]-sized batch.
usage: lw_check.py w [w ...]"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from seqwin_amd.device import Batch

ng, rpg, rl, k = 512, 50, 96000, 21
b = Batch.synthetic(ng, rpg, rl, n_ancestors=5, snp_ppm=10000, seed=20260821)
tar = np.arange(ng) < ng // 2
for w in [int(x) for x in sys.argv[1:]] or (200, 1024, 4096, 4097, 10000, 90000, 95980, 95981):
    ix = b.build_index(k, w, tar)
    best = None
    for _ in range(3):
        ix2 = b.build_index(k, w, tar)
        t = ix2.timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
        ix2.close()
    t = best
    print(f"split={os.environ.get('SEQWIN_AMD_WINDOW_SPLIT', '-')} w={w}: sizes {ix.sizes()} checksums {[hex(c)[-6:] for c in ix.checksums()]} total {t['total_ms']:.2f} ms "
          f"(sketch {t['sketch_ms']:.2f}, order {t['order_ms']:.2f}, nodes {t['nodes_ms']:.2f}, edges {t['edges_ms']:.2f})", flush=True)
    ix.close()
