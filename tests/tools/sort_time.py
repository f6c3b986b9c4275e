"""Time sw_sort_keys64 on n million random 54-bit keys (the edge-pair sort of a 15k build).  usage: sort_time.py [n_million] [bits]"""
import ctypes, os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from seqwin_amd._lib import c_u64, c_vp, check, lib
n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 745_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 54
g = torch.Generator(device="cuda").manual_seed(1)
keys = torch.randint(0, 2**bits, (n,), dtype=torch.int64, device="cuda", generator=g)
a, b = keys.clone(), torch.empty_like(keys)
best = 1e9
for _ in range(3):
    a.copy_(keys)
    flag, ms = ctypes.c_int(), ctypes.c_double()
    check(lib.sw_sort_keys64(c_vp(a.data_ptr()), c_vp(b.data_ptr()), c_u64(n), c_u64(0), c_u64(bits), c_vp(0), ctypes.byref(flag), ctypes.byref(ms)))
    best = min(best, ms.value)
tag = " ".join(f"{k[11:]}={v}" for k, v in os.environ.items() if k.startswith("SEQWIN_AMD_"))
print(f"{tag or 'default':50s} n={n} bits={bits}: {best:.2f} ms", flush=True)
