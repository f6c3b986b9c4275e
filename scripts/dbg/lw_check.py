"""Sanity of the route for windows above SW_MAX_WINDOW: sizes and time on a configs[1]-sized batch."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from seqwin_amd.device import Batch

ng, rpg, rl, k = 512, 50, 96000, 21
b = Batch.synthetic(ng, rpg, rl, n_ancestors=5, snp_ppm=10000, seed=20260821)
tar = np.arange(ng) < ng // 2
for w in (200, 1024, 4096, 4097, 10000, 90000, 95980, 95981):
    t0 = time.time()
    ix = b.build_index(k, w, tar)
    t1 = time.time()
    ix2 = b.build_index(k, w, tar)
    t = ix2.timings()
    print(f"w={w}: sizes {ix.sizes()} first {1e3 * (t1 - t0):.1f} ms, again total {t['total_ms']:.2f} ms (sketch {t['sketch_ms']:.2f}, order {t['order_ms']:.2f}, "
          f"nodes {t['nodes_ms']:.2f}, edges {t['edges_ms']:.2f}) expected density 2/(w+1) -> {int(ng * rpg * (rl - k + 1 - w + 1) * 2 / (w + 1))}", flush=True)
    ix.close(); ix2.close()
