"""Sketch / whole-path rate as a function of contig length (same 2.4576 Gbp, k=21, w=200): short records fill their tile badly."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from seqwin_amd.device import Batch
G = 512
for rpg, rl in ((50, 96_000), (200, 24_000), (600, 8_000), (1000, 4_800), (2400, 2_000), (4800, 1_000)):
    b = Batch.synthetic(G, rpg, rl, n_ancestors=5, snp_ppm=10_000, seed=7)
    tar = np.arange(G) % 2 == 0
    best = None
    for it in range(3):
        ix = b.build_index(21, 200, tar)
        t = ix.timings()
        if best is None or t["total_ms"] < best["total_ms"]:
            best = dict(t)
        sizes = ix.sizes()
        ix.close()
    bp = G * rpg * rl
    print(f"contigs of {rl:6d} bp: sketch {best['sketch_ms']:7.3f} ms ({bp / best['sketch_ms'] / 1e6:6.0f} Gbp/s), total {best['total_ms']:7.3f} ms "
          f"({bp / best['total_ms'] / 1e6:6.0f} Gbp/s), tiles {int(best['n_tiles'])}, kmers {sizes[0]}", flush=True)
    del b
