"""CPU: the oracle (oracle/seqwin_oracle.c) against every golden vector the path has.

* the reference's own golden graph (tests/smoke/fixtures/expected/graph.npz, k=17 w=10), copied as data;
* vectors produced by the compiled reference (tests/golden/make_golden.py): smoke FASTA at five (k, w),
  a synthetic pan-genome, edge-case FASTA (N runs, lowercase, U, IUPAC, empty / short records, CRLF,
  duplicates, gzip, empty file), w=1 hash known answers for 14 values of k, and the operator-level
  get_penalty / filter_kmers cases of the reference's tests/smoke/test_graph.py.
"""
import numpy as np
import pytest

import oracle
from conftest import GOLDEN, assert_graph_equal, load_case


def test_reference_golden_graph(smoke_paths):
    exp = np.load(GOLDEN / "smoke" / "expected_graph_k17_w10.npz")
    got = oracle.build(smoke_paths, 17, 10)
    assert_graph_equal(got, exp)
    assert len(got[0]) == 1061 and len(got[1]) == 380 and len(got[2]) == 404
    w, c = np.unique(got[2]["weight"], return_counts=True)
    assert dict(zip(w.tolist(), c.tolist())) == {1: 67, 2: 162, 3: 34, 4: 141}


def test_all_reference_vectors(manifest):
    assert len(manifest["cases"]) >= 28
    for case in manifest["cases"]:
        paths, z = load_case(case)
        got = oracle.build(paths, case["k"], case["w"])
        assert_graph_equal(got, z, case["ids"])
        if case["is_targets"] is not None and len(got[1]):
            nodes = got[1].copy()
            oracle.get_penalty(got[0], nodes, got[3], case["is_targets"])
            assert np.array_equal(nodes, z["nodes_scored"]), case["name"]


def test_operator_vectors():
    z = np.load(GOLDEN / "vectors" / "operators.npz")
    nodes = z["pen_nodes"].copy()
    oracle.get_penalty(z["pen_kmers"], nodes, z["pen_offsets"], z["pen_targets"])
    assert np.array_equal(nodes, z["pen_scored"])
    assert np.array_equal(nodes["n_tar"], [2, 0, 0, 1, 0, 0]) and np.array_equal(nodes["n_neg"], [1, 1, 1, 0, 0, 2])
    k2, n2 = oracle.filter_kmers(z["flt_kmers"], z["flt_nodes"], z["flt_used"])
    assert np.array_equal(k2, z["flt_kmers_out"]) and np.array_equal(n2, z["flt_nodes_out"])


def _srol(x):
    m = ((x & 0x8000000000000000) >> 30) | ((x & 0x100000000) >> 32)
    return ((x << 1) & 0xFFFFFFFDFFFFFFFF) | m


def _hash_by_definition(kmer: str, k: int):
    """F = XOR srol^{k-1-i} S[s_i]; R = XOR srol^{i} S[comp s_i]; SURVEY section 7 rule 3."""
    S = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    f = r = 0
    for i, c in enumerate(kmer):
        x = S[c]
        for _ in range(k - 1 - i):
            x = _srol(x)
        f ^= x
        y = S[comp[c]]
        for _ in range(i):
            y = _srol(y)
        r ^= y
    mh = (f + r) & 0xFFFFFFFFFFFFFFFF
    t = (mh * (1 ^ ((k * 0x90b45d39fb6da1fa) & 0xFFFFFFFFFFFFFFFF))) & 0xFFFFFFFFFFFFFFFF
    return mh, t ^ (t >> 27)


@pytest.mark.parametrize("k", [3, 4, 5, 16, 21, 31, 32, 33, 34, 62, 66])
def test_nthash_by_definition(k):
    rng = np.random.default_rng(k)
    seq = "".join(rng.choice(list("ACGT"), 120))
    seq = seq[:50] + "N" + seq[51:]
    mh, oh, pos = oracle.nthash(seq, k)
    exp_pos = [p for p in range(len(seq) - k + 1) if "N" not in seq[p:p + k]]
    assert pos.tolist() == exp_pos
    for i in (0, 1, len(pos) // 2, len(pos) - 1):
        e = _hash_by_definition(seq[int(pos[i]):int(pos[i]) + k], k)
        assert (int(mh[i]), int(oh[i])) == e


def test_minimizer_window_rule():
    """Brute force of SURVEY section 7 rule 4 against the ring-buffer restatement."""
    rng = np.random.default_rng(5)
    for _ in range(30):
        n = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTN"), n, p=[.24, .24, .24, .24, .04]))
        k = int(rng.integers(3, 12)); w = int(rng.integers(1, 30))
        mh, oh, pos = oracle.nthash(seq, k)
        exp = []
        last = -1
        for i in range(w - 1, len(mh)):
            win = mh[i - w + 1:i + 1]
            j = i - w + 1 + (len(win) - 1 - int(np.argmin(win[::-1])))   # rightmost minimum
            if int(pos[j]) > last and int(mh[j]) != 2**64 - 1:
                last = int(pos[j]); exp.append((int(oh[j]), int(pos[j])))
        if k > len(seq) or w > len(seq) - k + 1:
            exp = []
        _, goh, gpos = oracle.minimize(seq, k, w)
        assert list(zip(goh.tolist(), gpos.tolist())) == exp


def test_argument_errors(tmp_path):
    with pytest.raises(ValueError):
        oracle.build([], 2, 10)
    with pytest.raises(ValueError):
        oracle.build([], 21, 0)
    with pytest.raises(RuntimeError, match="Unable to open FASTA"):
        oracle.build([tmp_path / "missing.fa"], 21, 200)
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>r\nACGT\n")
    with pytest.raises(RuntimeError, match="sequence encountered before header"):
        oracle.build([bad], 21, 200)
    ctl = tmp_path / "ctl.fa"
    ctl.write_bytes(b">r\nACGT\x01ACGT\n")
    with pytest.raises(ValueError, match="control byte"):
        oracle.build([ctl], 3, 1)


def test_oracle_under_sanitizers(tmp_path):
    """The C restatement rebuilt with AddressSanitizer + UBSan (`make -C oracle asan`) reproduces the golden vectors and a
    ragged fuzz set without a sanitizer report.  CPU only."""
    import os
    import subprocess
    import sys
    root = GOLDEN.parent.parent
    if subprocess.run(["make", "-C", str(root / "oracle"), "asan"], capture_output=True).returncode != 0:
        pytest.skip("no sanitizer runtime for gcc here")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan):
        pytest.skip("libasan.so not found")
    script = r'''
import json, sys, random
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle
from conftest import GOLDEN, load_case
man = json.loads((GOLDEN / "manifest.json").read_text())
for case in man["cases"]:
    paths, z = load_case(case)
    k, n, e, o, ids = oracle.build(paths, case["k"], case["w"])
    assert np.array_equal(k, z["kmers"]) and np.array_equal(n, z["nodes"]) and np.array_equal(e, z["edges"]), case["name"]
    if case["is_targets"] is not None and len(n):
        oracle.get_penalty(k, n, o, case["is_targets"])
        assert np.array_equal(n, z["nodes_scored"])
        oracle.filter_kmers(k, n, frozenset(np.uint64(h) for h in n["hash"][::2]))
rng = random.Random(3)
for it in range(25):
    p = sys.argv[2] + f"/f{it}.fa"
    with open(p, "w") as f:
        for r in range(rng.randrange(0, 4)):
            f.write(f">r{r}\n" + "".join(rng.choice("ACGTNacgtRY") for _ in range(rng.choice([0, 3, 40, 700, 6000]))) + "\n")
    oracle.build([p], rng.choice([3, 7, 21, 33]), rng.choice([1, 5, 50, 200]))
print("SAN_OK")
'''
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
               SEQWIN_ORACLE_LIB=str(root / "oracle" / "libseqwin_oracle_asan.so"))
    out = subprocess.run([sys.executable, "-c", script, str(root), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "SAN_OK" in out.stdout, out.stderr[-3000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]


def test_checksums_change_when_any_field_moves_between_elements():
    """seqwin_amd.device.host_checksums (= csrc/device.hpp ck_kmer / ck_node / ck_edge, what the full-size tests pin the 75 Gbp
    arrays with): EVERY field of a node / edge enters with the element's index, the penalty by bit pattern -- swapping two nodes'
    counts, starts, stops or penalties, or two edges' seconds or weights, changes the sum (until r05 those
    terms were index-free sums, so the full-size arrays were pinned only up to permutation in them).  Whole rows are what the
    reference's own tests compare (tests/smoke/test_graph.py:281-291).  The shares of slices still add up modulo 2^64."""
    from seqwin_amd.device import host_checksums
    paths = sorted((GOLDEN / "synth").glob("pan_*.fa"))
    k, n, e, ro, _ = oracle.build(paths, 15, 20)
    oracle.get_penalty(k, n, ro, [i % 2 == 0 for i in range(len(paths))])
    base = host_checksums(k, n, e)
    differing = lambda arr, f: next((i, j) for i in range(len(arr)) for j in range(i + 1, min(i + 50, len(arr))) if arr[f][i] != arr[f][j])
    for field in ("hash", "start", "stop", "n_tar", "n_neg", "penalty"):
        i, j = differing(n, field)
        m = n.copy()
        m[field][[i, j]] = m[field][[j, i]]
        got = host_checksums(k, m, e)
        assert got[1] != base[1] and got[0] == base[0] and got[2] == base[2], field
    for field in ("first", "second", "weight"):
        i, j = differing(e, field)
        m = e.copy()
        m[field][[i, j]] = m[field][[j, i]]
        assert host_checksums(k, n, m)[2] != base[2], field
    m = k.copy()
    m[[3, 4]] = m[[4, 3]]
    assert host_checksums(m, n, e)[0] != base[0]
    # the penalty enters by bit pattern: -0.0 is not 0.0
    m = n.copy()
    z = int(np.flatnonzero(m["penalty"] == 0.0)[0]) if np.any(m["penalty"] == 0.0) else 0
    m["penalty"][z] = 0.0
    b0 = host_checksums(k, m, e)[1]
    m["penalty"][z] = -0.0
    assert host_checksums(k, m, e)[1] != b0
    # shares of slices add up
    ck, cn, ce = len(k) // 3, len(n) // 2, len(e) // 4
    parts = [host_checksums(k[:ck], n[:cn], e[:ce]), host_checksums(k[ck:], n[cn:], e[ce:], ck, cn, ce)]
    assert tuple(sum(p[i] for p in parts) % 2**64 for i in range(3)) == base
