"""Where does a fuzz case with a window above SW_MAX_WINDOW spend its time?  (GPU build vs oracle)"""
import os, random, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import oracle
from seqwin_amd import _core
tmp = tempfile.mkdtemp(prefix="lwt_")
rng = random.Random(5)
for it in range(24):
    ps = []
    for a in range(3):
        txt = []
        for r in range(3):
            L = rng.choice([400, 9000, 30000, 70000])
            if it % 3 == 0:
                unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 2, 7, 64])))
                s = (unit * (L // len(unit) + 1))[:L]
            else:
                s = "".join(rng.choice("ACGT") for _ in range(L))
            txt.append(f">r{r}\n{s}\n")
        p = os.path.join(tmp, f"{it}_{a}.fa")
        open(p, "w").write("".join(txt))
        ps.append(p)
    k = rng.choice([15, 21, 31])
    w = [200, 4096, 4097, 20000][it % 4]
    t0 = time.time()
    got = _core._build_native(ps, k, w, 1, False)
    t1 = time.time()
    exp = oracle.build(ps, k, w)
    t2 = time.time()
    ok = all(np.array_equal(x, y) for x, y in zip(got[:4], exp[:4]))
    print(f"case {it} lowcx={it % 3 == 0} k={k} w={w}: gpu {1e3 * (t1 - t0):.1f} ms, oracle {1e3 * (t2 - t1):.1f} ms, kmers {len(got[0])} ok={ok}", flush=True)
