"""Cost of issuing the tuple-exchange collectives (RCCL, world size 1) on top of the sharded build."""
import os, sys, time
os.environ["SEQWIN_DIST_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from seqwin_amd import dist as swdist
from seqwin_amd.device import Batch, set_device
set_device(0)
G = 512
batch = Batch.synthetic(G, 50, 96000, n_ancestors=5, snp_ppm=10000, seed=20260821)
tar = np.arange(G) % 2 == 0
shard = swdist.Shard(batch, first_assembly=0, n_assemblies_total=G)
eng = swdist.HipEngine("device")
ix = None
for it in range(8):
    if ix is not None: ix.merged.close() if hasattr(ix, "merged") else ix.close()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = swdist.build_sharded_index(shard, 21, 200, tar, engine=eng)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    tm = ix.timings()
    print(f"step {it}: {dt*1e3:.2f} ms", {k: round(v, 2) for k, v in tm.items() if k.endswith('wall_ms')}, flush=True)
dist.destroy_process_group()
