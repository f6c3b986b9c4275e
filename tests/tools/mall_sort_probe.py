"""Does a radix pass over data that fits the 256 MiB Infinity Cache move fewer bytes past the L2 (FETCH_SIZE / WRITE_SIZE), and
is it faster per element?  (a node sort whose first pass is MSD into <= 100 MB buckets and whose remaining
passes run bucket by bucket inside the Infinity Cache.)

Sorts pairs (u32 key + 16-byte payload, csrc/radix.hip's pair passes through sw_sort_pairs32) of several sizes:
  * n = 2.9 M pairs: one top-byte bucket of the 745 M occurrences of the 15 000-genome build -- 58 MB of elements, 116 MB with the
    second buffer of the pass: resident;
  * n = 23 M: 8 buckets, 0.46 GB + 0.46 GB: not resident;
  * n = 186 M: a quarter of the build.
Per size: `reps` sorts of 24 key bits (the three passes that would follow an MSD pass) back to back on the same buffers; the
time per sort comes from the library's own events.  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs) the
per-kernel counters tell whether the resident case is served on-die as far as those counters are concerned.

    python3 tests/tools/mall_sort_probe.py [reps]
"""
import ctypes, os, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from seqwin_amd._lib import check, lib

os.environ["SEQWIN_AMD_SORT"] = "own"
os.environ["SEQWIN_AMD_PAIR_SORT"] = "own"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = torch.Generator(device="cuda").manual_seed(11)
for n in (2_900_000, 23_000_000, 186_000_000):
    keys = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
    vals = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    k, ka, v, va = keys.clone(), torch.empty_like(keys), vals, torch.empty_like(vals)
    times = []
    for r in range(reps):
        flag, ms = ctypes.c_int(), ctypes.c_double()
        check(lib.sw_sort_pairs32(k.data_ptr(), ka.data_ptr(), v.data_ptr(), va.data_ptr(), n, 24, None, ctypes.byref(flag), ctypes.byref(ms)))
        times.append(ms.value)
        if flag.value:
            k, ka, v, va = ka, k, va, v
    torch.cuda.synchronize()
    best = min(times[1:])
    print(f"n = {n:>11,d} pairs ({n * 20 / 1e6:8.1f} MB of elements): 3 passes {best:8.3f} ms = {best / 3 * 1e3:8.1f} us per pass, "
          f"{3 * 2 * n * 20 / best / 1e6:7.1f} GB/s of moved bytes, {best / n * 745.1e6:7.2f} ms per 745 M elements", flush=True)
