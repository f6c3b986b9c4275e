"""CPU, dev container only: the oracle against the compiled reference (oracle/_ref) on seeded fuzz.

Skipped where oracle/_ref has not been built (it can only be built where /root/reference exists)."""
import gzip
import random

import numpy as np
import pytest

import oracle

ref = oracle.load_ref()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs /root/reference)")


def _randseq(rng, n):
    mode = rng.random()
    s = []
    for _ in range(n):
        r = rng.random()
        if r < 0.01:
            s.append(rng.choice("NnRYKMxX-*"))
        elif r < 0.03:
            s.append(rng.choice("acgtuU"))
        else:
            s.append(rng.choice("ACGT" if mode < 0.8 else "AC"))
    s = "".join(s)
    if rng.random() < 0.3:
        p = rng.randrange(0, max(1, n))
        s = s[:p] + "N" * rng.randrange(1, 60) + s[p:]
    return s


def test_fuzz_build_matches_reference(tmp_path):
    rng = random.Random(1)
    for it in range(60):
        ps = []
        for a in range(rng.randrange(1, 5)):
            txt = ""
            for r in range(rng.randrange(0, 4)):
                s = _randseq(rng, rng.choice([0, 5, 30, 100, 400, 1500]))
                txt += f">r{r} desc\n"
                width = rng.choice([60, 80, 7])
                for i in range(0, len(s), width):
                    txt += s[i:i + width] + rng.choice(["\n", "\r\n", " \n"])
                if rng.random() < 0.2:
                    txt += "\n  \n"
            gz = rng.random() < 0.3
            p = tmp_path / (f"{it}_{a}.fa" + (".gz" if gz else ""))
            if gz:
                with gzip.open(p, "wt") as f:
                    f.write(txt)
            else:
                p.write_text(txt)
            ps.append(str(p))
        k = rng.choice([3, 4, 5, 7, 15, 16, 17, 18, 19, 21, 31, 32, 33, 40])
        w = rng.choice([1, 2, 3, 5, 10, 25, 50, 200])
        a = oracle.build(ps, k, w)
        b = ref._build_native(ps, k, w, rng.choice([1, 2, 3]), rng.random() < 0.5)
        for x, y in zip(a[:4], b[:4]):
            assert x.dtype == y.dtype and np.array_equal(x, y)
        assert list(a[4]) == list(b[4])
        if len(a[1]) and len(ps) >= 2:
            tar = [i % 2 == 0 for i in range(len(ps))]
            n1, n2 = a[1].copy(), b[1].copy()
            oracle.get_penalty(a[0], n1, a[3], tar)
            ref._get_penalty_native(b[0], n2, b[3], np.asarray(tar, np.bool_), 2)
            assert np.array_equal(n1, n2)
            used = set(int(h) for h in n1["hash"][::3])
            f1 = oracle.filter_kmers(a[0], n1, used)
            f2 = ref._filter_kmers_native(b[0], n2, list(used))
            assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])


# The GPU parity tests lean on the oracle for k up to 300 and windows up to 10^6 (tests/test_gpu_parity.py); the fuzz
# above stops at k = 40, w = 200.  The same comparison on the wide range: every k the GPU tests use against the oracle
# (64, 100, 255, 256, 257, 300) plus 41 .. 1000, windows 1000, 4097, 20 000, 100 000, records with N runs, three assemblies.
WIDE_K = [41, 47, 63, 64, 65, 100, 128, 200, 255, 256, 257, 300, 511, 512, 1000]
WIDE_W = [1, 2, 31, 200, 1000, 4097, 20_000, 100_000]


@pytest.mark.parametrize("k", WIDE_K)
def test_wide_k_and_w_match_reference(tmp_path, k):
    rng = random.Random(1000 + k)
    ps = []
    for a in range(3):
        txt = ""
        for r in range(rng.randrange(1, 4)):
            n = rng.choice([k - 1, k, k + 1, 3000, 12_000, 45_000, 130_000])
            seq = "".join(rng.choice("ACGT") for _ in range(max(0, n)))
            if n > 2000 and rng.random() < 0.7:       # N runs, some of them closer than k to each other
                for _ in range(rng.randrange(1, 5)):
                    p = rng.randrange(0, n)
                    seq = seq[:p] + "N" * rng.choice([1, 2, k // 2 + 1, 500]) + seq[p:]
            if rng.random() < 0.3:
                seq = seq.lower().replace("t", "u")
            txt += f">a{a}r{r}\n{seq}\n"
        if a == 2:                                    # a contig shared with assembly 0: nodes and edges with weight 2
            txt += ps_first_contig
        p = tmp_path / f"w{k}_{a}.fa"
        p.write_text(txt)
        if a == 0:
            ps_first_contig = txt[:txt.index("\n", txt.index("\n") + 1) + 1]
        ps.append(str(p))
    tar = [True, False, True]
    for w in WIDE_W:
        a = oracle.build(ps, k, w)
        b = ref._build_native(ps, k, w, 2, False)
        for x, y in zip(a[:4], b[:4]):
            assert x.dtype == y.dtype and np.array_equal(x, y), (k, w)
        if len(a[1]):
            n1, n2 = a[1].copy(), b[1].copy()
            oracle.get_penalty(a[0], n1, a[3], tar)
            ref._get_penalty_native(b[0], n2, b[3], np.asarray(tar, np.bool_), 2)
            assert np.array_equal(n1, n2), (k, w)
