"""Long differential fuzz on the GPU box: HIP path vs oracle on random FASTA sets and (k, w).
usage: python tests/tools/fuzz_gpu.py SECONDS [SEED]"""
import gzip, os, random, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle
from seqwin_amd import _core

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
replay = [int(x) for x in sys.argv[3:]]
LOWMEM = os.environ.get("FUZZ_LOWMEM") == "1"      # sw_build's chunked low-memory route (set SEQWIN_AMD_LOWMEM_CHUNK_MBP=0: one assembly per chunk)
DIST = os.environ.get("FUZZ_DIST") == "1"          # also push every case through the routed multi-GPU forms (P shards on one GPU)
GZ_ALL = os.environ.get("FUZZ_GZ") == "1"          # every file gzipped (levels 0-9): with SEQWIN_AMD_DEVICE_INFLATE=1 the device ingest route
if DIST:
    sys.path.insert(0, str(ROOT / "tests"))
    from test_gpu_dist import _route_and_merge, routed_tuple_exchange          # optional: replay these exact case seeds and print the first differences
tmp = tempfile.mkdtemp(prefix="fuzz_")
t_end = time.time() + budget
n_cases = n_bad = 0
it = int(os.environ.get("FUZZ_START", "0"))   # (continue a campaign's seed sequence from its it-th case)
while time.time() < t_end and (not replay or it < len(replay)):
    seed = replay[it] if replay else seed0 * 1000003 + it
    it += 1
    rng = random.Random(seed)
    style = rng.random()
    ps = []
    for a in range(rng.randrange(1, 5)):
        txt = []
        for r in range(rng.randrange(0, 4)):
            L = rng.choice([0, 3, 20, 33, 64, 100, 230, 400, 1500, 8191, 8192, 8193, 9000, 17000, 30000, 70000])
            mode = rng.random()
            if style < 0.15:      # low complexity / repeats
                unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 2, 3, 5, 7, 31, 64, 200])))
                s = (unit * (L // len(unit) + 1))[:L]
                s = "".join(c if rng.random() > 0.002 else rng.choice("ACGT") for c in s)
            else:
                alpha = "ACGT" if mode < 0.8 else "AC"
                s = "".join(rng.choice(alpha) for _ in range(L))
            if rng.random() < 0.5 and L:
                arr = list(s)
                for _ in range(rng.randrange(0, 6)):
                    p = rng.randrange(0, L)
                    for j in range(p, min(L, p + rng.choice([1, 1, 2, 5, 40, 300]))):
                        arr[j] = rng.choice("NnRYKMxX-*")
                s = "".join(arr)
            if rng.random() < 0.2:
                s = "".join(c.lower() if rng.random() < 0.3 else c for c in s).replace("t", rng.choice("tuU"))
            txt.append(f">r{r} x\n")
            width = rng.choice([60, 80, 7, 100000])
            for i in range(0, len(s), width):
                txt.append(s[i:i + width] + rng.choice(["\n", "\r\n", " \n"]))
        gz = rng.random() < 0.2 or GZ_ALL
        p = os.path.join(tmp, f"{it}_{a}.fa" + (".gz" if gz else ""))
        data = "".join(txt)
        if gz:
            with gzip.open(p, "wt", compresslevel=(seed + a) % 10 if GZ_ALL else 9) as f:
                f.write(data)
        else:
            with open(p, "w") as f:
                f.write(data)
        ps.append(p)
    k = rng.choice([3, 4, 5, 7, 11, 15, 16, 17, 19, 21, 31, 32, 33, 47, 64, 65, 100, 255, 256, 257])
    w = rng.choice([1, 2, 3, 5, 10, 15, 16, 17, 25, 31, 32, 33, 34, 50, 63, 64, 65, 100, 200, 201, 500, 1000, 4096, 4097, 5000, 20000])
    if os.environ.get("FUZZ_TRACE"):   # the case that is about to run, for a process that does not come back
        with open(os.environ["FUZZ_TRACE"] + f".{os.getpid()}", "a") as tf:
            tf.write(f"{it} seed={seed} k={k} w={w} files={[os.path.getsize(p) for p in ps]}\n")
    try:
        t_a = time.time()
        got = _core._build_native(ps, k, w, rng.choice([1, 3]), LOWMEM)
        t_b = time.time()
        exp = oracle.build(ps, k, w)
        if os.environ.get("FUZZ_SLOW") and time.time() - t_a > float(os.environ["FUZZ_SLOW"]):
            print(f"slow case seed={seed} k={k} w={w}: build {t_b - t_a:.2f} s, oracle {time.time() - t_b:.2f} s, "
                  f"kmers {len(exp[0])}, files {[os.path.getsize(p) for p in ps]}", flush=True)
        ok = all(np.array_equal(x, y) for x, y in zip(got[:4], exp[:4])) and [tuple(t) for t in got[4]] == [tuple(t) for t in exp[4]]
        if ok and len(got[1]) and len(ps) >= 2:
            tar = [i % 2 == 0 for i in range(len(ps))]
            oracle.get_penalty(exp[0], exp[1], exp[3], tar)
            _core._get_penalty_native(got[0], got[1], got[3], np.asarray(tar, np.bool_), 1)
            ok = np.array_equal(got[1], exp[1])
        if ok and DIST and len(ps) >= 2:
            tar = [i % 2 == 0 for i in range(len(ps))]
            P = rng.choice([2, 3, 4, 8])
            for fn in (routed_tuple_exchange, _route_and_merge):
                kw = {"requests": seed % 2 == 1} if fn is routed_tuple_exchange else {}   # hashes by request on every other case
                d = fn(ps, P, k, w, tar, **kw)
                ok = ok and np.array_equal(d[0], exp[0]) and np.array_equal(d[1], exp[1]) and np.array_equal(d[2], exp[2]) \
                    and np.array_equal(d[3], exp[3])
                if not ok:
                    print("  dist form", fn.__name__, "P =", P)
                    if True:
                        d2 = fn(ps, P, k, w, tar, **kw)
                        print("    retry equal to expected:", all(np.array_equal(a, b) for a, b in zip(d2[:4], exp[:4])),
                              " retry equal to first:", all(np.array_equal(a, b) for a, b in zip(d2[:4], d[:4])))
                        for name, x, y in zip(("kmers", "nodes", "edges", "offsets"), d[:4], exp[:4]):
                            print("    ", name, len(x), len(y))
                            if len(x) == len(y) and not np.array_equal(x, y):
                                j = np.nonzero(x != y)[0]
                                print("       ", len(j), "diffs; first at", j[:4], x[j[:4]], y[j[:4]])
                    break
    except Exception as e:
        ok = False
        print("EXC", type(e).__name__, e)
    n_cases += 1
    if not ok:
        n_bad += 1
        print(f"MISMATCH seed={seed} k={k} w={w} files={ps}")
        if replay:
            for name, x, y in zip(("kmers", "nodes", "edges", "offsets"), got[:4], exp[:4]):
                print("  ", name, len(x), len(y))
                if len(x) == len(y) and not np.array_equal(x, y):
                    d = np.nonzero(x != y)[0][:3]
                    print("     first diffs at", d, x[d], y[d])
            for pth in ps:
                recs = oracle.read_fasta(pth)
                print("   file", pth, [(i, len(sq), sq.count(b"N")) for i, sq in recs])
            oh = sorted(zip(exp[0]["record_idx"].tolist(), exp[0]["pos"].tolist()))
            gh = sorted(zip(got[0]["record_idx"].tolist(), got[0]["pos"].tolist()))
            print("   only in expected:", sorted(set(oh) - set(gh))[:10], " only in got:", sorted(set(gh) - set(oh))[:10])
    for p in ps:
        os.unlink(p)
try:   # SEQWIN_AMD_POOL_DEBUG soaks: what the pool saw
    import ctypes
    from seqwin_amd._lib import lib as _lib
    _st = (ctypes.c_uint64 * 3)()
    _lib.sw_pool_debug_stats(_st)
    print(f"pool: debug mode {'on' if _st[0] else 'off'}, {_st[1]} blocks written after their release, {_st[2]} host-side hand-overs between threads / streams")
except Exception as e:
    print("pool stats unavailable:", e)
print(f"fuzz: {n_cases} cases, {n_bad} mismatches, seed0={seed0}")
sys.exit(1 if n_bad else 0)
