"""First GPU contact: parity of the sketch stream and of the full build against the oracle."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle
from seqwin_amd import _core
from seqwin_amd.device import Batch, host_checksums

G = ROOT / "tests/golden"
man = json.loads((G / "manifest.json").read_text())
bad = 0
for case in man["cases"]:
    paths = [str(G / p) for p in case["paths"]]
    z = np.load(G / "vectors" / f"{case['name']}.npz")
    try:
        k, n, e, o, ids = _core._build_native(paths, case["k"], case["w"], 2, False)
        ok = (np.array_equal(k, z["kmers"]) and np.array_equal(n, z["nodes"]) and np.array_equal(e, z["edges"])
              and np.array_equal(o, z["record_offsets"]) and [list(t) for t in ids] == case["ids"])
        if ok and case["is_targets"] is not None and len(n):
            _core._get_penalty_native(k, n, o, np.asarray(case["is_targets"], np.bool_), 1)
            ok = np.array_equal(n, z["nodes_scored"])
            if not ok:
                d = np.nonzero(n != z["nodes_scored"])[0][:5]
                print("  penalty mismatch", n[d], z["nodes_scored"][d])
    except Exception as ex:
        ok = False
        print("  EXC", type(ex).__name__, ex)
    print(case["name"], "OK" if ok else "MISMATCH", len(z["kmers"]), len(z["nodes"]), len(z["edges"]))
    if not ok:
        bad += 1
        if 'k' in dir() and len(k) != len(z["kmers"]):
            print("   n_kmers", len(k), "vs", len(z["kmers"]))

# synthetic batch: sketch stream vs oracle, full index vs oracle (through FASTA written from the batch)
import tempfile, os
for (ng, rpg, rl, k, w) in [(6, 3, 20000, 21, 200), (4, 2, 50000, 15, 50), (3, 1, 100000, 31, 200), (2, 2, 3000, 17, 10)]:
    b = Batch.synthetic(ng, rpg, rl, n_ancestors=2, snp_ppm=10000, seed=7)
    t0 = time.time(); oh, km = b.sketch(k, w); t1 = time.time()
    eh, ep, er = [], [], []
    tmp = tempfile.mkdtemp()
    paths = []
    offs, ids = b.records()
    for a in range(ng):
        p = os.path.join(tmp, f"a{a}.fa")
        with open(p, "w") as f:
            for r in range(int(offs[a]), int(offs[a + 1])):
                seq = b.record(r)
                f.write(f">{ids[a][r - int(offs[a])]}\n{seq.decode()}\n")
                mh, o2, pos = oracle.minimize(seq, k, w)
                eh.append(o2); ep.append(pos.astype(np.uint32)); er.append(np.full(len(pos), r, np.uint32))
        paths.append(p)
    eh = np.concatenate(eh); ep = np.concatenate(ep); er = np.concatenate(er)
    ok = len(oh) == len(eh) and np.array_equal(oh, eh) and np.array_equal(km["pos"], ep) and np.array_equal(km["record_idx"], er)
    print(f"sketch synth {ng}x{rpg}x{rl} k{k} w{w}: {'OK' if ok else 'MISMATCH'} n={len(oh)} exp={len(eh)} ({t1-t0:.3f}s)")
    bad += not ok
    tar = [i < ng // 2 or i == 0 for i in range(ng)]
    ix = b.build_index(k, w, tar)
    K, N, E = ix.export()
    ek, en, ee, eo, eids = oracle.build(paths, k, w)
    oracle.get_penalty(ek, en, eo, tar)
    ok = np.array_equal(K, ek) and np.array_equal(N, en) and np.array_equal(E, ee)
    print(f"index  synth: {'OK' if ok else 'MISMATCH'} kmers {len(K)}/{len(ek)} nodes {len(N)}/{len(en)} edges {len(E)}/{len(ee)}", ix.timings())
    if ok:
        assert ix.checksums() == host_checksums(ek, en, ee), (ix.checksums(), host_checksums(ek, en, ee))
    else:
        for name, x, y in (("kmers", K, ek), ("nodes", N, en), ("edges", E, ee)):
            if len(x) == len(y):
                d = np.nonzero(x != y)[0]
                if len(d): print("   first diff", name, d[:3], x[d[:3]], y[d[:3]])
    bad += not ok
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
