"""The keys-only radix sort of csrc/radix.hip against torch.sort (stable) and against rocPRIM, with timings.
usage: python tests/tools/sort_check.py [n_million]"""
import ctypes, os, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from seqwin_amd._lib import c_u64, c_vp, check, lib

def sort(keys, begin, end):
    a, b = keys.clone(), torch.empty_like(keys)
    flag, ms = ctypes.c_int(), ctypes.c_double()
    check(lib.sw_sort_keys64(c_vp(a.data_ptr()), c_vp(b.data_ptr()), c_u64(a.numel()), c_u64(begin), c_u64(end), c_vp(0),
                             ctypes.byref(flag), ctypes.byref(ms)))
    return (b if flag.value else a), ms.value

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 50_000_000
g = torch.Generator(device="cuda").manual_seed(1)
for name, begin, end, make in [
    ("random 64 bits", 0, 64, lambda: torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device="cuda", generator=g)),
    ("pairs 54 bits", 0, 54, lambda: torch.randint(0, 2**54, (n,), dtype=torch.int64, device="cuda", generator=g)),
    ("few digits 5 bits @ 13", 13, 18, lambda: torch.randint(0, 2**40, (n,), dtype=torch.int64, device="cuda", generator=g)),
    ("unsort words bits [46, 62)", 46, 62, lambda: (torch.randperm(n, device="cuda", generator=g) << 32) | torch.arange(n, device="cuda")),
    ("one digit value", 0, 16, lambda: torch.full((n,), 0x1234, dtype=torch.int64, device="cuda") + (torch.arange(n, device="cuda") << 20)),
]:
    keys = make()
    for impl in ("own atomic", "own ballot", "rocprim"):
        if impl == "rocprim":
            os.environ["SEQWIN_AMD_SORT"] = "rocprim"
        else:
            os.environ["SEQWIN_AMD_SORT"] = "own"    # (below 2^26 keys the library would take rocPRIM by itself)
            os.environ["SEQWIN_AMD_RADIX_RANK"] = impl.split()[1]
        out, ms = sort(keys, begin, end)
        out, ms = sort(keys, begin, end)
        # expected: stable sort by the masked key field
        field = (keys >> begin) & ((1 << (end - begin)) - 1) if end - begin < 64 else keys
        if end - begin == 64:   # unsigned order of int64 bit patterns
            field = keys ^ (-2**63)
        order = torch.sort(field, stable=True).indices
        ok = bool(torch.equal(out, keys[order]))
        print(f"{name:32s} {impl:11s} n={n}: {ms:8.3f} ms  {'OK' if ok else 'MISMATCH'}", flush=True)
        assert ok
for small in (0, 1, 63, 64, 65, 4095, 4096, 4097, 100_001):
    keys = torch.randint(0, 2**30, (small,), dtype=torch.int64, device="cuda", generator=g)
    os.environ["SEQWIN_AMD_SORT"] = "own"
    os.environ["SEQWIN_AMD_RADIX_RANK"] = "atomic"
    out, _ = sort(keys, 3, 27)
    field = (keys >> 3) & ((1 << 24) - 1)
    assert torch.equal(out, keys[torch.sort(field, stable=True).indices]), small
print("small sizes OK")
