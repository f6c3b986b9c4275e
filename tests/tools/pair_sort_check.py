"""The pair passes of csrc/radix.hip (32-bit keys + 16-byte payload, the node sort) against torch.sort (stable) and against
rocPRIM, with timings.  usage: python tests/tools/pair_sort_check.py [n_million] [time_n_million]"""
import ctypes, os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from seqwin_amd._lib import c_u64, c_vp, check, lib


def sort(keys, vals, end_bit):
    k, ka = keys.clone(), torch.empty_like(keys)
    v, va = vals.clone(), torch.empty_like(vals)
    flag, ms = ctypes.c_int(), ctypes.c_double()
    check(lib.sw_sort_pairs32(c_vp(k.data_ptr()), c_vp(ka.data_ptr()), c_vp(v.data_ptr()), c_vp(va.data_ptr()), c_u64(k.numel()),
                              c_u64(end_bit), c_vp(0), ctypes.byref(flag), ctypes.byref(ms)))
    return (ka, va, ms.value) if flag.value else (k, v, ms.value)


n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 20_000_000
g = torch.Generator(device="cuda").manual_seed(7)
bad = 0
cases = [("random 32 bits", 32, lambda m: torch.randint(-2**31, 2**31 - 1, (m,), dtype=torch.int32, device="cuda", generator=g)),
         ("few values (long runs)", 32, lambda m: torch.randint(0, 1000, (m,), dtype=torch.int32, device="cuda", generator=g) * 4_000_037),
         ("16 key bits", 16, lambda m: torch.randint(0, 2**31 - 1, (m,), dtype=torch.int32, device="cuda", generator=g)),
         ("one value", 32, lambda m: torch.full((m,), 0x12345678, dtype=torch.int32, device="cuda")),
         ("descending", 32, lambda m: torch.arange(m, 0, -1, dtype=torch.int32, device="cuda") * 64)]
for name, end_bit, make in cases:
    for m in (1, 5, 4096, 4097, 70_001, n):
        keys = make(m)
        vals = torch.stack([torch.arange(m, device="cuda", dtype=torch.int32)] * 4, dim=1).contiguous()   # payload = original index x 4
        vals[:, 1] ^= 0x5A5A5A5A
        vals[:, 2] += 7
        vals[:, 3] = keys
        field = (keys.to(torch.int64) & 0xFFFFFFFF) & ((1 << end_bit) - 1)
        order = torch.sort(field, stable=True).indices
        for impl in ("own", "rocprim"):
            os.environ["SEQWIN_AMD_SORT"] = impl
            os.environ["SEQWIN_AMD_PAIR_SORT"] = impl      # (the pair passes of radix.hip are taken only on request)
            for rank in (("atomic",) if impl == "own" else ("-",)):       # (the pair passes rank by LDS atomics only)
                os.environ["SEQWIN_AMD_RADIX_RANK"] = rank
                print(f"  .. {name} {impl} n={m}", flush=True)
                k, v, ms = sort(keys, vals, end_bit)
                ok = bool(torch.equal(k, keys[order])) and bool(torch.equal(v, vals[order]))
                bad += not ok
                if m == n or not ok:
                    print(f"{name:26s} {impl:8s} rank={rank:7s} n={m}: {ms:8.3f} ms  {'OK' if ok else 'MISMATCH'}", flush=True)
print("pair sort:", "all OK" if not bad else f"{bad} MISMATCHES", flush=True)
if len(sys.argv) > 2:       # timing at the size of the 15k build
    m = int(float(sys.argv[2]) * 1e6)
    keys = torch.randint(-2**31, 2**31 - 1, (m,), dtype=torch.int32, device="cuda", generator=g)
    vals = torch.zeros((m, 4), dtype=torch.int32, device="cuda")
    for impl, rank in (("own", "atomic"), ("rocprim", "-")):
        os.environ["SEQWIN_AMD_SORT"], os.environ["SEQWIN_AMD_RADIX_RANK"], os.environ["SEQWIN_AMD_PAIR_SORT"] = impl, rank, impl
        best = min(sort(keys, vals, 32)[2] for _ in range(3))
        print(f"timing n={m} 32 bits {impl:8s} rank={rank:7s}: {best:.2f} ms", flush=True)
sys.exit(1 if bad else 0)
