"""Does RCCL at world size 1 return what was sent?  all_to_all_single / all_gather_into_tensor on large device tensors:
(A) one collective at a time, (B) with an asynchronous all_gather still in flight when the next collective is issued
(what seqwin_amd.dist.build_sharded_index did until round 3)."""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 149_000_000
for mode in ("A: serial", "B: async all_gather in flight"):
    for rep in range(2):
        x = torch.randint(0, 2**62, (n,), dtype=torch.int64, device=dev)
        c = torch.randint(0, 2**62, (n // 2, 2), dtype=torch.int64, device=dev)
        h = torch.randint(0, 2**62, (n // 4,), dtype=torch.int64, device=dev)
        table = torch.empty_like(h); work = dist.all_gather_into_tensor(table, h, async_op=True)
        if mode.startswith("A"):
            work.wait()
        ox = torch.empty_like(x); dist.all_to_all_single(ox, x, [n], [n])
        oc = torch.empty_like(c); dist.all_to_all_single(oc, c, [n // 2], [n // 2])
        work.wait()
        torch.cuda.synchronize()
        print(f"{mode} rep {rep}: keys ({n * 8 >> 20} MiB) {torch.equal(ox, x)} cand {torch.equal(oc, c)} table {torch.equal(table, h)}"
              f" first mismatch keys {int((ox != x).nonzero()[0]) if not torch.equal(ox, x) else -1}", flush=True)
dist.destroy_process_group()
