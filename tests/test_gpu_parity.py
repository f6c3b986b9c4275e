"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle and the golden vectors.

Structured like the reference's tests/smoke/test_graph.py (same helper names and assertions where the
reference has a test), plus golden-vector, fuzz and full-size property tests.  Integer / index results
are compared bit-for-bit; the one floating-point field (penalty, f64) is also required to be
bit-identical (tolerance 0) because the reference's own thread-invariance test compares whole node
rows with np.array_equal (test_graph.py:281-291).
"""
import gzip
import os
import random
from pathlib import Path

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, assert_graph_equal, load_case
from seqwin_amd import EDGE_DTYPE, KMER_DTYPE, NODE_DTYPE, KmerGraph, _filter_kmers, _get_penalty
from seqwin_amd.device import Batch, host_checksums

pytestmark = pytest.mark.gpu


def _build(*args, **kwargs):
    graph = KmerGraph(*args, **kwargs)
    return graph.kmers, graph.nodes, graph.edges, graph.record_offsets, graph.record_ids


def _sorted_edges(edges: np.ndarray) -> np.ndarray:
    edge_values = edges.view(np.uint64).reshape(-1, 3)
    idx = np.lexsort((edge_values[:, 2], edge_values[:, 1], edge_values[:, 0]))
    return edge_values[idx]


def _assert_node_ranges(kmers: np.ndarray, nodes: np.ndarray) -> None:
    assert np.all(nodes["start"] <= nodes["stop"]) and (len(nodes) == 0 or nodes["stop"][-1] == len(kmers))
    assert np.array_equal(nodes["start"][1:], nodes["stop"][:-1])
    if len(nodes):
        assert nodes["start"][0] == 0
    rec = kmers["record_idx"].astype(np.int64)
    inner = np.ones(len(kmers), bool)
    inner[nodes["start"][nodes["start"] < len(kmers)]] = False
    assert np.all((np.diff(rec) >= 0) | ~inner[1:])
    assert int((nodes["stop"] - nodes["start"]).sum()) == len(kmers)


# ---- golden vectors ---------------------------------------------------------------------------------

def test_reference_golden_graph(smoke_paths):
    exp = np.load(GOLDEN / "smoke" / "expected_graph_k17_w10.npz")
    assert_graph_equal(_build(smoke_paths, 17, 10, n_cpu=1), exp)
    assert_graph_equal(_build(smoke_paths, 17, 10, n_cpu=2, low_memory=True), exp)


def test_all_reference_vectors(manifest):
    for case in manifest["cases"]:
        paths, z = load_case(case)
        got = _build(paths, case["k"], case["w"], n_cpu=2)
        assert_graph_equal(got, z, case["ids"])
        if case["is_targets"] is not None and len(got[1]):
            assert _get_penalty(got[0], got[1], got[3], case["is_targets"]) is None
            assert np.array_equal(got[1], z["nodes_scored"]), case["name"]   # penalty bit-exact (tolerance 0)


def test_operator_vectors():
    z = np.load(GOLDEN / "vectors" / "operators.npz")
    nodes = z["pen_nodes"].copy()
    _get_penalty(z["pen_kmers"], nodes, z["pen_offsets"], z["pen_targets"])
    assert np.array_equal(nodes, z["pen_scored"])
    k2, n2 = _filter_kmers(z["flt_kmers"], z["flt_nodes"], frozenset(np.uint64(h) for h in z["flt_used"]))
    assert np.array_equal(k2, z["flt_kmers_out"]) and np.array_equal(n2, z["flt_nodes_out"])


# ---- mirrors of the reference's test_graph.py ---------------------------------------------------------

def test_build_threading_equivalence(smoke_paths):   # test_graph.py:67-127
    a = _build(smoke_paths, kmerlen=7, windowsize=10, n_cpu=1)
    b = _build(smoke_paths, kmerlen=7, windowsize=10, n_cpu=2)
    c = _build(smoke_paths, kmerlen=7, windowsize=10, n_cpu=99)
    kmers_1, nodes_1, edges_1, offs_1, ids_1 = a
    assert kmers_1.dtype == KMER_DTYPE and nodes_1.dtype == NODE_DTYPE and edges_1.dtype == EDGE_DTYPE
    assert offs_1.dtype == np.uint32 and np.array_equal(offs_1, [0, 1, 2, 3, 4])
    assert np.array_equal(np.unique(kmers_1["record_idx"]), np.arange(4, dtype=np.uint32))
    assert np.all(nodes_1["n_tar"] == 0) and np.all(nodes_1["n_neg"] == 0) and np.all(nodes_1["penalty"] == 0.0)
    ev = edges_1.view(np.uint64).reshape(-1, 3)
    assert np.array_equal(ev[:, 0], edges_1["first"]) and np.array_equal(ev[:, 2], edges_1["weight"])
    assert nodes_1.flags.writeable and kmers_1.flags.c_contiguous
    for g in (a, b, c):
        _assert_node_ranges(g[0], g[1])
    for other in (b, c):
        assert np.array_equal(kmers_1, other[0]) and np.array_equal(nodes_1, other[1])
        assert np.array_equal(_sorted_edges(edges_1), _sorted_edges(other[2]))
        assert np.array_equal(offs_1, other[3]) and ids_1 == other[4]
    assert len(ids_1) == 4
    assert_graph_equal(a, dict(zip(("kmers", "nodes", "edges", "record_offsets"), oracle.build(smoke_paths, 7, 10)[:4])))


def test_multi_record_offsets_and_global_record_indices(tmp_path: Path):   # test_graph.py:144-165
    seq = "ACGT" * 20
    paths = []
    for i, n_records in enumerate([2, 1, 3, 1]):
        p = tmp_path / f"a{i}.fasta"
        p.write_text("".join(f">r{j}\n{seq}\n" for j in range(n_records)))
        paths.append(p)
    kmers, _, _, offs, ids = _build(paths, kmerlen=7, windowsize=10, n_cpu=2)
    assert [len(t) for t in ids] == [2, 1, 3, 1]
    assert np.array_equal(offs, np.array([0, 2, 3, 6, 7], dtype=np.uint32))
    assert np.array_equal(np.unique(kmers["record_idx"]), np.arange(7, dtype=np.uint32))


def test_build_empty_record_offsets(tmp_path: Path):   # test_graph.py:168-187
    empty = tmp_path / "empty.fasta"
    empty.write_text("")
    for paths, expected in (([], [0]), ([empty], [0, 0])):
        for low_memory in (False, True):
            kmers, nodes, edges, offs, ids = _build(paths, kmerlen=7, windowsize=10, n_cpu=2, low_memory=low_memory)
            assert len(kmers) == 0 and len(nodes) == 0 and len(edges) == 0
            assert offs.dtype == np.uint32 and np.array_equal(offs, expected)
            assert ids == [()] * len(paths)


@pytest.mark.parametrize("chunk_mbp", [None, "0", "1"])
def test_low_memory_build_matches_standard(smoke_paths, tmp_path, monkeypatch, chunk_mbp):   # test_graph.py:222-245
    """low_memory streams the assemblies through HBM in chunks (sw_build: build_chunked) and indexes the concatenated
    tuple stream: one assembly per chunk (SEQWIN_AMD_LOWMEM_CHUNK_MBP=0), a few per chunk (1 Mbp), or the default
    chunk size; also chosen automatically when the files exceed SEQWIN_AMD_HBM_BUDGET_GB."""
    if chunk_mbp is not None:
        monkeypatch.setenv("SEQWIN_AMD_LOWMEM_CHUNK_MBP", chunk_mbp)
    for n_cpu in (1, 2, 99):
        standard = _build(smoke_paths, kmerlen=7, windowsize=10, n_cpu=n_cpu, low_memory=False)
        low = _build(smoke_paths, kmerlen=7, windowsize=10, n_cpu=n_cpu, low_memory=True)
        for a, b in zip(standard[:4], low[:4]):
            assert np.array_equal(a, b)
        assert standard[4] == low[4]
    # ragged inputs: multi-record assemblies, an empty file, records shorter than k, gz, N runs; chunks of ~1 Mbp
    rng = np.random.default_rng(5)
    paths = []
    for a in range(9):
        recs = []
        for r in range(int(rng.integers(0, 4))):
            n = int(rng.choice([0, 12, 5000, 300_000, 700_000]))
            seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n, p=[0.2475, 0.2475, 0.2475, 0.2475, 0.01]).tobytes()
            recs.append(b">a%d_r%d x\n" % (a, r) + seq + b"\n")
        p = tmp_path / (f"lm{a}.fa" + (".gz" if a % 4 == 3 else ""))
        if a % 4 == 3:
            with gzip.open(p, "wb") as f:
                f.write(b"".join(recs))
        else:
            p.write_bytes(b"".join(recs))
        paths.append(p)
    for k, w in ((21, 200), (15, 10)):
        exp = oracle.build(paths, k, w)
        got = _build(paths, k, w, n_cpu=3, low_memory=True)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])
    exp = dict(zip(("kmers", "nodes", "edges", "record_offsets"), oracle.build(paths, 21, 200)[:4]))
    import logging
    for budget, streamed in (("1", False), ("0.001", True)):   # the ~4 MB of files fit 1 GB (one-shot) but not 1 MB (chunks)
        monkeypatch.setenv("SEQWIN_AMD_HBM_BUDGET_GB", budget)
        seen = []
        h = logging.Handler()
        h.emit = lambda rec: seen.append(rec.getMessage())
        logging.getLogger().addHandler(h)
        old = logging.getLogger().level
        logging.getLogger().setLevel(logging.INFO)
        try:
            assert_graph_equal(_build(paths, 21, 200), exp)
        finally:
            logging.getLogger().removeHandler(h)
            logging.getLogger().setLevel(old)
        assert any("streamed through HBM in chunks" in m for m in seen) == streamed


def test_filter_kmers():   # test_graph.py:190-219
    kmers = np.array([(10, 0), (11, 0), (20, 1), (30, 2), (31, 2), (32, 2)], dtype=KMER_DTYPE)
    nodes = np.array([(10, 0, 2, 1, 0, 0.1), (20, 2, 3, 1, 0, 0.2), (30, 3, 6, 1, 1, 0.3)], dtype=NODE_DTYPE)
    kmers_new, nodes_new = _filter_kmers(kmers, nodes, {30, 10})
    assert np.array_equal(nodes_new["hash"], np.array([10, 30], dtype=np.uint64))
    assert np.array_equal(nodes_new["start"], [0, 2]) and np.array_equal(nodes_new["stop"], [2, 5])
    assert np.array_equal(nodes_new["penalty"], [0.1, 0.3]) and np.array_equal(nodes_new["n_neg"], [0, 1])
    assert np.array_equal(kmers_new, np.array([(10, 0), (11, 0), (30, 2), (31, 2), (32, 2)], dtype=KMER_DTYPE))
    e = _filter_kmers(kmers, nodes, set())
    assert len(e[0]) == 0 and len(e[1]) == 0
    e = _filter_kmers(kmers, nodes, [99, 30, 30])
    assert np.array_equal(e[1]["hash"], [30]) and len(e[0]) == 3


def _synthetic_penalty_inputs():   # test_graph.py:248-266
    kmers = np.array([(0, 0), (1, 0), (2, 1), (3, 2), (4, 4), (5, 2), (6, 3), (7, 5), (8, 6), (9, 4)], dtype=KMER_DTYPE)
    nodes = np.array([(10, 0, 5, 0, 0, 0.0), (20, 5, 7, 0, 0, 0.0), (30, 7, 9, 0, 0, 0.0), (40, 9, 10, 0, 0, 0.0),
                      (50, 10, 10, 9, 9, 9.0), (60, 5, 9, 0, 0, 0.0)], dtype=NODE_DTYPE)
    return kmers, nodes, np.array([0, 2, 4, 5, 7], dtype=np.uint32), np.array([True, False, True, False])


def test_get_penalty_exact_scoring():   # test_graph.py:269-278
    kmers, nodes, offs, tar = _synthetic_penalty_inputs()
    assert _get_penalty(kmers, nodes, offs, tar, n_cpu=1) is None
    assert np.array_equal(nodes["n_tar"], [2, 0, 0, 1, 0, 0]) and np.array_equal(nodes["n_neg"], [1, 1, 1, 0, 0, 2])
    np.testing.assert_allclose(nodes["penalty"], [0.5, np.hypot(1.0, 0.5), np.hypot(1.0, 0.5), 0.5, 1.0, np.sqrt(2.0)])
    ref = _synthetic_penalty_inputs()[1]
    oracle.get_penalty(kmers, ref, offs, tar)
    assert np.array_equal(nodes, ref)
    many = _synthetic_penalty_inputs()[1]
    _get_penalty(kmers, many, offs, tar, n_cpu=99)   # test_graph.py:281-291
    assert np.array_equal(nodes, many)


def test_get_penalty_skips_zero_record_assemblies():   # test_graph.py:294-304
    kmers = np.array([(0, 0), (1, 1)], dtype=KMER_DTYPE)
    nodes = np.array([(10, 0, 2, 0, 0, 0.0)], dtype=NODE_DTYPE)
    _get_penalty(kmers, nodes, np.array([0, 1, 1, 1, 2], dtype=np.uint32), [True, False, True, False], n_cpu=2)
    assert nodes[0]["n_tar"] == 1 and nodes[0]["n_neg"] == 1 and nodes[0]["penalty"] == np.sqrt(0.5)


def test_get_penalty_validation():   # test_graph.py:307-341 (device-side checks; host-side ones are in test_abi_cpu.py)
    kmers, nodes, offs, tar = _synthetic_penalty_inputs()
    bad_nodes = nodes.copy(); bad_nodes[0]["stop"] = len(kmers) + 1
    with pytest.raises(ValueError):
        _get_penalty(kmers, bad_nodes, offs, tar)
    bad_kmers = kmers.copy(); bad_kmers[0]["record_idx"] = 7
    with pytest.raises(ValueError):
        _get_penalty(bad_kmers, nodes.copy(), offs, tar)
    descending = kmers.copy(); descending[3]["record_idx"] = 0
    with pytest.raises(ValueError):
        _get_penalty(descending, nodes.copy(), offs, tar)


def test_penalty_f64_bit_exact_over_all_counts():
    """Every (n_tar, n_neg) pair for several (T, NT): the device's mul/add/sqrt sequence must round like x86-64."""
    for T, NT in [(1, 1), (3, 7), (72, 99), (256, 256), (13, 1000)]:
        A = T + NT
        # node i has one occurrence in each of the first a targets and first b non-targets
        combos = [(a, b) for a in range(T + 1) for b in range(NT + 1) if a + b > 0]
        if len(combos) > 30000:
            combos = combos[::len(combos) // 30000 + 1]
        recs, starts = [], [0]
        for a, b in combos:
            recs += list(range(a)) + list(range(T, T + b))
            starts.append(len(recs))
        kmers = np.zeros(len(recs), KMER_DTYPE); kmers["record_idx"] = recs
        nodes = np.zeros(len(combos), NODE_DTYPE)
        nodes["hash"] = np.arange(len(combos)); nodes["start"] = starts[:-1]; nodes["stop"] = starts[1:]
        offs = np.arange(A + 1, dtype=np.uint32)
        tar = np.array([True] * T + [False] * NT)
        exp = nodes.copy()
        oracle.get_penalty(kmers, exp, offs, tar)
        _get_penalty(kmers, nodes, offs, tar)
        assert np.array_equal(nodes["n_tar"], [c[0] for c in combos]) and np.array_equal(nodes["n_neg"], [c[1] for c in combos])
        assert np.array_equal(nodes["penalty"].view(np.uint64), exp["penalty"].view(np.uint64)), (T, NT)


# ---- fuzz against the oracle --------------------------------------------------------------------------

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
        s = s[:p] + "N" * rng.randrange(1, 300) + s[p:]
    return s


def test_fuzz_build_matches_oracle(tmp_path):
    rng = random.Random(2)
    for it in range(60):
        ps = []
        for a in range(rng.randrange(1, 5)):
            txt = ""
            for r in range(rng.randrange(0, 4)):
                s = _randseq(rng, rng.choice([0, 5, 30, 100, 400, 1500, 9000, 30000]))
                txt += f">r{r} desc\n"
                width = rng.choice([60, 80, 7, 1000])
                for i in range(0, len(s), width):
                    txt += s[i:i + width] + rng.choice(["\n", "\r\n", " \n"])
            gz = rng.random() < 0.3
            p = tmp_path / (f"{it}_{a}.fa" + (".gz" if gz else ""))
            if gz:
                with gzip.open(p, "wt") as f:
                    f.write(txt)
            else:
                p.write_text(txt)
            ps.append(p)
        k = rng.choice([3, 4, 5, 7, 15, 16, 17, 18, 19, 21, 31, 32, 33, 40, 100, 255, 256, 257, 300])
        w = rng.choice([1, 2, 3, 5, 10, 15, 16, 17, 25, 31, 32, 33, 34, 50, 66, 200, 1000, 4096, 4097, 6000, 20000, 10**6])
        got = _build(ps, k, w, n_cpu=rng.choice([1, 3]))
        exp = oracle.build(ps, k, w)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])
        if len(got[1]) and len(ps) >= 2:
            tar = [i % 2 == 0 for i in range(len(ps))]
            oracle.get_penalty(exp[0], exp[1], exp[3], tar)
            _get_penalty(got[0], got[1], got[3], tar)
            assert np.array_equal(got[1], exp[1])
            used = frozenset(np.uint64(h) for h in exp[1]["hash"][::3])
            f1 = _filter_kmers(got[0], got[1], used); f2 = oracle.filter_kmers(exp[0], exp[1], used)
            assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])


def test_low_complexity_and_ties(tmp_path):
    """Homopolymers / tandem repeats: every window is a tie, so the rightmost-minimum rule is all that decides."""
    p = tmp_path / "lc.fa"
    p.write_text(">polyA\n" + "A" * 20000 + "\n>at\n" + "AT" * 9000 + "\n>rep7\n" + "ACGGTCA" * 3000 + "\n>mix\n" +
                 "A" * 500 + "N" + "C" * 700 + "ACGT" * 300 + "\n")
    for k, w in [(21, 200), (5, 33), (15, 10), (31, 1000), (21, 4097), (5, 8000), (15, 17990)]:
        got = _build([p], k, w)
        exp = oracle.build([p], k, w)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])))


@pytest.mark.parametrize("split", [None, "4096,1024", "16,5"], ids=["default", "direct-to-4096", "two-step-above-16"])
def test_tile_seams_and_many_gaps(tmp_path, monkeypatch, split):
    """Records spanning many tiles, with valid stretches of every length around k (idx space != pos space); windows up to
    the tile limit taken directly, and the two-step route for longer ones (default from SW_WINDOW_SPLIT on)."""
    if split:
        monkeypatch.setenv("SEQWIN_AMD_WINDOW_SPLIT", split)
    rng = np.random.default_rng(11)
    s = "".join(rng.choice(list("ACGT"), 120000))
    cuts = sorted(rng.choice(len(s), 400, replace=False).tolist())
    arr = list(s)
    for c in cuts:
        for j in range(c, min(len(arr), c + int(rng.integers(1, 40)))):
            arr[j] = "N"
    gap = "".join(arr)
    p = tmp_path / "seams.fa"
    p.write_text(f">long\n{s}\n>gappy\n{gap}\n>dense_gaps\n" + "N".join(s[i:i + 23] for i in range(0, 60000, 23)) + "\n")
    for k, w in [(21, 200), (21, 50), (23, 7), (24, 3), (17, 2048), (19, 2049), (17, 4096), (17, 4097), (21, 30000), (23, 100000), (15, 119986), (15, 119987)]:
        got = _build([p], k, w)
        exp = oracle.build([p], k, w)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])))


# ---- device-resident pipeline ---------------------------------------------------------------------------

def _oracle_for_batch(b: Batch, k, w, tar, tmp_path):
    offs, ids = b.records()
    paths = []
    for a in range(len(offs) - 1):
        p = tmp_path / f"syn{a}.fa"
        with open(p, "wb") as f:
            for r in range(int(offs[a]), int(offs[a + 1])):
                f.write(b">" + ids[a][r - int(offs[a])].encode() + b"\n" + b.record(r) + b"\n")
        paths.append(p)
    exp = oracle.build(paths, k, w)
    if tar is not None:
        oracle.get_penalty(exp[0], exp[1], exp[3], tar)
    return exp


@pytest.mark.parametrize("shape,k,w", [((12, 5, 60000), 21, 200), ((5, 2, 150000), 15, 200), ((4, 1, 200000), 31, 200),
                                       ((6, 3, 20000), 19, 10), ((4, 1, 200000), 21, 5000), ((4, 2, 150000), 15, 100000)])
def test_synthetic_batch_index_matches_oracle(tmp_path, shape, k, w):
    ng, rpg, rl = shape
    b = Batch.synthetic(ng, rpg, rl, n_ancestors=3, snp_ppm=10000, seed=20260821)
    tar = [i < ng // 2 for i in range(ng)]
    exp = _oracle_for_batch(b, k, w, tar, tmp_path)
    oh, km = b.sketch(k, w)
    order = np.lexsort((exp[0]["pos"], exp[0]["record_idx"]))
    assert np.array_equal(km, exp[0][order])                      # the tuple stream in (record_idx, pos) order
    node_of = np.repeat(np.arange(len(exp[1])), (exp[1]["stop"] - exp[1]["start"]).astype(np.int64))
    assert np.array_equal(oh, exp[1]["hash"][node_of][order])
    ix = b.build_index(k, w, tar)
    K, N, E = ix.export()
    assert np.array_equal(K, exp[0]) and np.array_equal(N, exp[1]) and np.array_equal(E, exp[2])
    assert ix.checksums() == host_checksums(exp[0], exp[1], exp[2])
    t = ix.timings()
    assert t["total_bp"] == ng * rpg * rl and t["sketch_launches"] == 1
    # idempotence: a second build of the same batch is identical
    assert b.build_index(k, w, tar).checksums() == ix.checksums()


@pytest.mark.parametrize("k,w", [(21, 200), (15, 10), (31, 50)])
def test_ragged_assemblies_with_scaffold_gaps_match_oracle(tmp_path, monkeypatch, k, w):
    """The shape of real draft assemblies (sw_batch_synthetic_ragged; reference inputs: tests/targets.txt): per genome 20-300 contigs
    of 200 bp ... 1.5 Mbp, scaffold gaps of 10-1000 N in one contig of ten -- 64-thread tiles for the short contigs, gap tiles and
    the generic kernel's list mode next to the 256-thread tiles, in one batch.  From the device batch, and from its FASTA text (1 % of
    the bases in lower case) through the host ingest: kmers / scored nodes / edges of the oracle (N-skipping:
    nthash_kmer.hpp:315-333, 491-511; windows span the gaps: minimizer.cpp:69-70)."""
    b = Batch.synthetic_ragged(7, 150_000, n_ancestors=2, snp_ppm=10000, seed=20260821)
    info = b.info()
    assert info["n_assemblies"] == 7 and info["n_records"] >= 7 * 20
    lens = [len(b.record(r)) for r in range(info["n_records"])]
    assert min(lens) >= 200 and max(lens) <= 1_500_000 and sum(lens) == info["total_bp"]
    assert any(b"N" in b.record(r) for r in range(info["n_records"]))
    tar = [i % 2 == 0 for i in range(7)]
    exp = _oracle_for_batch(b, k, w, tar, tmp_path)
    ix = b.build_index(k, w, tar)
    K, N, E = ix.export()
    assert np.array_equal(K, exp[0]) and np.array_equal(N, exp[1]) and np.array_equal(E, exp[2])
    t = ix.timings()
    assert t["tiles_b256"] and t["tiles_b64"] and t["tiles_gap"], t          # every tile class of the plan is in play
    assert ix.checksums() == host_checksums(exp[0], exp[1], exp[2])
    # the same genomes as FASTA text with soft-masked bases, through sw_build
    from bench import write_fasta_fast
    monkeypatch.setenv("SEQWIN_AMD_WRITE_LOWER_PPM", "10000")
    d = tmp_path / "fa"
    d.mkdir()
    paths, bp = write_fasta_fast(b, 7, str(d), 4)
    assert bp == info["total_bp"] and any(c.islower() for c in open(paths[0]).read(200000) if c.isalpha())
    g = KmerGraph(paths, kmerlen=k, windowsize=w, n_cpu=3)
    _get_penalty(g.kmers, g.nodes, g.record_offsets, tar)
    assert np.array_equal(g.kmers, exp[0]) and np.array_equal(g.nodes, exp[1]) and np.array_equal(g.edges, exp[2])
    # a shard of the job holds the same genomes (bench.py --gpus N)
    s = Batch.synthetic_ragged(3, 150_000, n_ancestors=2, snp_ppm=10000, seed=20260821, first_genome=4)
    off = b.record_offsets()
    assert [s.record(r) for r in range(s.info()["n_records"])] == [b.record(r) for r in range(int(off[4]), int(off[7]))]


def test_window_split_route_at_full_size(monkeypatch):
    """Windows above SW_MAX_WINDOW are sketched with a smaller window and selected from that superset (index.hip:
    select in order_tuples).  The test knob sends w = 200 down that route on a configs[1]-sized batch -- sketch with
    w' = 16, 2.9e8 candidates, a five-level tree of minima -- and the arrays must be those of the direct route."""
    ng, rpg, rl, k, w = 512, 50, 96000, 21, 200
    tar = np.arange(ng) < ng // 2
    b = Batch.synthetic(ng, rpg, rl, n_ancestors=5, snp_ppm=10000, seed=20260821)
    ix = b.build_index(k, w, tar)
    want, sizes = ix.checksums(), ix.sizes()
    ix.close()
    b.close()
    monkeypatch.setenv("SEQWIN_AMD_WINDOW_SPLIT", "64,16")
    b2 = Batch.synthetic(ng, rpg, rl, n_ancestors=5, snp_ppm=10000, seed=20260821)   # (plans are cached per batch)
    ix2 = b2.build_index(k, w, tar)
    assert ix2.sizes() == sizes and ix2.checksums() == want
    oh, km = b2.sketch(k, w)
    assert len(oh) == sizes[0] == len(km)


def test_full_size_properties():
    """configs[1]-sized synthetic batch (2.4 Gbp): size-independent properties + shard-sum identities."""
    ng, rpg, rl, k, w = 512, 50, 96000, 21, 200
    b = Batch.synthetic(ng, rpg, rl, n_ancestors=5, snp_ppm=10000, seed=20260821)
    tar = np.arange(ng) < ng // 2
    ix = b.build_index(k, w, tar)
    K, N, E = ix.export()
    nk, nn, ne = ix.sizes()
    assert (nk, nn, ne) == (len(K), len(N), len(E))
    assert ix.checksums() == host_checksums(K, N, E)
    assert 0.0097 < nk / (ng * rpg * rl) < 0.0102                  # minimizer density ~ 2/(w+1)
    assert np.all(N["hash"][1:] > N["hash"][:-1])                  # nodes strictly sorted by hash
    _assert_node_ranges(K, N)
    ev = E.view(np.uint64).reshape(-1, 3)
    assert np.all(ev[:, 0] <= ev[:, 1])
    key = ev[:, 0].astype(object) * (1 << 64) + ev[:, 1].astype(object) if len(ev) < 1000 else None
    lex = np.lexsort((ev[:, 1], ev[:, 0]))
    assert np.array_equal(lex, np.arange(len(ev)))                 # edges sorted by (first, second), no duplicates
    assert np.all((ev[1:, 0] != ev[:-1, 0]) | (ev[1:, 1] != ev[:-1, 1]))
    assert np.all(np.isin(ev[:, 0], N["hash"])) and np.all(np.isin(ev[:, 1], N["hash"]))
    assert E["weight"].min() >= 1 and E["weight"].max() <= ng
    assert np.all(N["n_tar"] <= ng // 2) and np.all(N["n_neg"] <= ng - ng // 2) and np.all(N["n_tar"] + N["n_neg"] >= 1)
    # host recomputation of the counts from the exported occurrences (numpy, independent of both C paths)
    asm = K["record_idx"] // rpg
    first = np.ones(len(K), bool); first[1:] = asm[1:] != asm[:-1]; first[N["start"].astype(np.int64)] = True
    node_of = np.repeat(np.arange(len(N)), (N["stop"] - N["start"]).astype(np.int64))
    assert np.array_equal(np.bincount(node_of[first & tar[asm]], minlength=len(N)), N["n_tar"])
    assert np.array_equal(np.bincount(node_of[first & ~tar[asm]], minlength=len(N)), N["n_neg"])
    ft = N["n_tar"] * (1.0 / (ng // 2)); fn = N["n_neg"] * (1.0 / (ng - ng // 2))
    assert np.array_equal(N["penalty"], np.sqrt((1.0 - ft) * (1.0 - ft) + fn * fn))
    # sum of edge weights == number of distinct (pair, assembly) combinations among adjacent occurrences
    assert int(E["weight"].sum()) <= nk - ng * rpg


# ---- next rows, device-resident -------------------------------------------------------------

def test_device_resident_filter_pipeline_matches_reference_numpy(tmp_path, smoke_paths):
    """threshold sums, _filter_edges_and_nodes, filter_kmers and graph.npz without leaving HBM, against the
    numpy expressions of the reference's kmers.py:132-173, 426-429 applied to oracle arrays."""
    paths = sorted((GOLDEN / "synth").glob("pan_*.fa")) + smoke_paths
    tar = [i % 2 == 0 for i in range(len(paths))]
    b = Batch.from_fasta(paths, n_cpu=2)
    ix = b.build_index(15, 20, tar)
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    oracle.get_penalty(ek, en, eo, tar)
    nt, ng = en["n_tar"].astype(np.uint64), en["n_neg"].astype(np.uint64)
    assert ix.threshold_sums() == (int(nt.sum()), int((nt * nt).sum()), int((nt * ng).sum()))
    for th in (0.0, 1.0, 1.9, 2.5, 100.0):
        f = ix.filter_graph(th)
        _, fn, fe = f.export()
        e2 = ee[ee["weight"] > np.uintp(th)]                                   # kmers.py:152-153
        keep = np.unique(e2.view(np.uint64).reshape(-1, 3)[:, :2])              # kmers.py:157
        n2 = en[np.searchsorted(en["hash"], keep)]                              # kmers.py:158-160
        assert np.array_equal(fe, e2) and np.array_equal(fn, n2)
        used = frozenset(np.uint64(h) for h in n2["hash"][::2])
        g = ix.filter_kmers(f, used)
        gk, gn, _ = g.export()
        rk, rn = oracle.filter_kmers(ek, n2, used)
        assert np.array_equal(gk, rk) and np.array_equal(gn, rn)
    out = tmp_path / "graph.npz"
    raw = b.build_index(15, 20, None)
    raw.save_npz(out, b.records()[0])
    z = np.load(out)
    ek0, en0, ee0, eo0, _ = oracle.build(paths, 15, 20)
    assert sorted(z.files) == ["edges", "kmers", "nodes", "record_offsets"]
    assert np.array_equal(z["kmers"], ek0) and np.array_equal(z["nodes"], en0)
    assert np.array_equal(z["edges"], ee0) and np.array_equal(z["record_offsets"], eo0)


def test_staging_overflow_rerun(tmp_path):
    """Poly-A: every window emits, far above the 2/(w+1) density the staging buffer is sized for -> the sketch
    is re-run with the exact size (run_sketch); the result must still be exact."""
    p = tmp_path / "polyA.fa"
    p.write_text(">a\n" + "A" * 300000 + "\n>b\n" + "ACGT" * 50000 + "\n")
    for k, w in [(21, 200), (15, 33)]:
        got = _build([p], k, w)
        exp = oracle.build([p], k, w)
        assert len(exp[0]) > 250000
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])))


def test_records_with_gaps_use_the_fast_kernel_tile_by_tile(tmp_path):
    """A record with invalid bases is cut into fast-class tiles; only tiles whose reach crosses a gap are pre-listed for the
    generic kernel (get_plan: gap_list).  Gaps are put inside tiles, right at tile seams (7968 window ends per tile at
    w = 200), at the record's ends and closer together than k; a record that is mostly gaps stays in the generic class."""
    rng = np.random.default_rng(99)
    def seq(n):
        return rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
    a = seq(200000)
    for start, ln in [(0, 3), (5000, 1), (7968 + 199 - 5, 40), (2 * 7968 + 230, 700), (30000, 20), (30015, 2), (45000, 1500), (199990, 10)]:
        a[start:start + ln] = ord("N")
    b = seq(40000)
    b[::150] = ord("N")                       # a gap in every tile reach: generic class
    c = seq(33000)                            # short enough for the 64-thread tile class (18 tiles, one of them crosses the gap)
    c[16000:16005] = ord("n")
    p = tmp_path / "gaps.fa"
    p.write_bytes(b">a\n" + a.tobytes() + b"\n>b\n" + b.tobytes() + b"\n>c\n" + c.tobytes() + b"\n>d\n" + seq(25000).tobytes() + b"\n")
    for k, w in [(21, 200), (31, 64), (15, 20)]:
        exp = oracle.build([p], k, w)
        got = _build([p], k, w)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])))
    t = Batch.from_fasta([p]).build_index(21, 200).timings()
    assert 2 <= t["ovf_tiles"] < t["n_tiles"] // 2     # some tiles of a and c went through the list pass, most did not


@pytest.mark.parametrize("slot_cap", ["1", "40", "90"])
def test_tile_slots_and_shared_overflow_area(tmp_path, slot_cap, monkeypatch):
    """Every tile writes its tuples into its own stage slot; a tile with more winners than the slot holds takes a range
    of the shared overflow area (one atomic), and the pass is re-run with the exact size when that area is too small.
    SEQWIN_AMD_SLOT_CAP shrinks the slot so that all (1), most (40) or some (90 of ~80 expected) tiles overflow."""
    rng = np.random.default_rng(int(slot_cap))
    paths = []
    for i in range(3):
        p = tmp_path / f"g{i}.fa"
        seq = "".join(rng.choice(list("ACGT"), 150000 + 977 * i))
        p.write_text(f">a{i}\n{seq[:90001]}\n>b{i}\n{seq[90001:110000]}NNN{seq[110000:]}\n")
        paths.append(p)
    exp = oracle.build(paths, 21, 200)
    monkeypatch.setenv("SEQWIN_AMD_SLOT_CAP", slot_cap)
    got = _build(paths, 21, 200)
    assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])))
    t = Batch.from_fasta(paths).build_index(21, 200).timings()
    assert t["sketch_launches"] in (1, 2)   # 2: the first overflow area (4096 entries here) was outgrown and the pass re-run
    monkeypatch.delenv("SEQWIN_AMD_SLOT_CAP")
    assert Batch.from_fasta(paths).build_index(21, 200).timings()["sketch_launches"] == 1


@pytest.mark.parametrize("w", [4, 5, 7, 8, 9, 10, 15, 16, 17, 31, 32, 33])
def test_window_equal_to_run_length_and_overflow_tiles(tmp_path, w, monkeypatch):
    """w == L (4 / 8 / 16 / 32) is the corner where the window never reaches past the previous run, and the tiles the
    fast kernel hands to the generic kernel (more suffix records than it publishes) must still fit that kernel's
    geometry (run length <= w).  Found by tests/tools/fuzz_gpu.py; SEQWIN_AMD_RC forces the hand-over for most tiles.
    (r04: windows of 4 ... 15 take the fast kernel with runs of 4 / 8 -- bit phase per lane, four or eight lanes per word
    of the emit bitmap.)"""
    rng = np.random.default_rng(w)
    p = tmp_path / "long.fa"
    p.write_text(">a\n" + "".join(rng.choice(list("ACGT"), 70000)) + "\n>b\n" + "".join(rng.choice(list("ACGT"), 8193)) + "\n")
    exp = oracle.build([p], 21, w)
    for rc in (None, "2"):
        if rc is None:
            monkeypatch.delenv("SEQWIN_AMD_RC", raising=False)
        else:
            monkeypatch.setenv("SEQWIN_AMD_RC", rc)
        got = _build([p], 21, w)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])))
        if w >= 8:    # the hand-over really happens when forced, and only rarely otherwise (runs of 4 never hold more than two records)
            t = Batch.from_fasta([p]).build_index(21, w).timings()
            assert (t["ovf_tiles"] >= t["n_tiles"] // 2) if rc else (t["ovf_tiles"] <= max(1, t["n_tiles"] // 4))


@pytest.mark.parametrize("w", [4, 6, 8, 10, 13, 15])
def test_small_windows_fast_class_equals_generic_kernel_and_oracle(tmp_path, w, monkeypatch):
    """Windows below 16 through the fast kernel (runs of 8 for 8 <= w < 16, of 4 for 4 <= w < 8; r04) against the generic kernel
    they used to take (SEQWIN_AMD_SKETCH=nosmall) and against the oracle: long and short records (both tile classes), invalid
    bases (gap tiles go to the generic kernel's list mode), homopolymers and tandem repeats (every window a tie), several k
    (pair-table warm-up up to 32, the stepped one above), lower-case and IUPAC."""
    rng = np.random.default_rng(100 + w)
    def seq(n):
        return "".join(rng.choice(list("ACGT"), n))
    a = tmp_path / "a.fa"
    a.write_text(">long\n" + seq(60_000) + "\n>short1\n" + seq(700) + "\n>short2\n" + seq(90) + "\n>gaps\n" + seq(5000) + "N" * 7 + seq(3000)
                 + "NN" + seq(40) + "R" + seq(9000) + "\n>tiny\n" + seq(w + 20) + "\n")
    b = tmp_path / "b.fa"
    b.write_text(">polyA\n" + "A" * 9000 + "\n>at\n" + "AT" * 4000 + "\n>rep7\n" + "ACGGTCA" * 1500 + "\n>lower\n" + seq(12_000).lower() + "\n")
    for k in (5, 17, 21, 32, 33, 40):
        exp = oracle.build([a, b], k, w)
        monkeypatch.delenv("SEQWIN_AMD_SKETCH", raising=False)
        got = _build([a, b], k, w, n_cpu=2)
        assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])
        monkeypatch.setenv("SEQWIN_AMD_SKETCH", "nosmall")
        old = _build([a, b], k, w, n_cpu=2)
        assert all(np.array_equal(x, y) for x, y in zip(got[:4], old[:4])), (k, w)


def test_pipelined_and_plain_upload_give_the_same_batch(tmp_path, monkeypatch):
    """FASTA -> HBM goes through the pinned ring while later files are still being parsed (default) or in one piece
    after parsing; with one or many workers; both must hand the kernels the same records."""
    rng = np.random.default_rng(11)
    paths = []
    for f in range(9):
        parts = []
        for r in range(int(rng.integers(1, 4))):
            n = int(rng.choice([0, 17, 800, 40_000, 300_000]))
            seq = rng.choice(np.frombuffer(b"ACGTN", np.uint8), n, p=[0.245, 0.245, 0.245, 0.245, 0.02]).tobytes()
            parts.append(b">c%d\n" % r + b"\n".join(seq[i:i + 70] for i in range(0, n, 70)) + b"\n")
        p = tmp_path / f"s{f}.fa"
        p.write_bytes(b"".join(parts))
        paths.append(str(p))
    ref = None
    for stream_upload in (True, False):
        for n_cpu in (1, 5):
            if stream_upload:
                monkeypatch.delenv("SEQWIN_AMD_NO_STREAM_UPLOAD", raising=False)
            else:
                monkeypatch.setenv("SEQWIN_AMD_NO_STREAM_UPLOAD", "1")
            b = Batch.from_fasta(paths, n_cpu=n_cpu)
            offs, ids = b.records()
            recs = [b.record(r) for r in range(int(offs[-1]))]
            ix = b.build_index(15, 25, np.arange(len(paths)) % 2 == 0)
            got = (offs.tolist(), ids, recs, [a.tobytes() for a in ix.export()])
            ix.close(); b.close()
            if ref is None:
                ref = got
                ek, en, ee, eo, _ = oracle.build(paths, 15, 25)
                oracle.get_penalty(ek, en, eo, [i % 2 == 0 for i in range(len(paths))])
                assert got[3] == [ek.tobytes(), en.tobytes(), ee.tobytes()]
            else:
                assert got == ref, (stream_upload, n_cpu)


def test_native_log_line_reaches_the_root_logger(smoke_paths, caplog):
    """log_python (cpp/src/utils/logging.cpp:9-29) logs to Python's root logger from native code; so does the binding."""
    import logging
    with caplog.at_level(logging.INFO):
        KmerGraph(smoke_paths, kmerlen=21, windowsize=200)
    assert any("MI355X index" in r.getMessage() and "4 assemblies" in r.getMessage() for r in caplog.records)


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the fields the driver reads (metric, value, unit, n_gpus, steps, warmup,
    ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus roofline and -- at N = 1 --
    cpu_baseline.  Run on the small workload, with a 4-genome CPU sample."""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1",
                          "--cpu-sample-genomes", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip() == lines[0]     # ONE line and nothing else on stdout
    d = json.loads(lines[0])
    # ... also when RCCL makes a communicator (it prints a version banner to stdout; r05: the bench keeps stdout to itself)
    import os
    out2 = subprocess.run([sys.executable, str(root / "bench.py"), "--workload", "tiny", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                          capture_output=True, text=True, timeout=600,
                          env=dict(os.environ, SEQWIN_DIST_FORCE_COLLECTIVES="1", SEQWIN_BENCH_FORCE_DIST="1", MASTER_PORT="29547"))
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert out2.stdout.count("\n") == 1 and out2.stdout.startswith("{"), out2.stdout[:300]
    d2 = json.loads(out2.stdout)
    assert d2["dist"]["world"] == 1 and d2["dist"]["distinct_gpus"] == 1 and d2["dist"]["ranks"][0]["rank"] == 0 and d2["dist"]["collectives"] == "issued"
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "plan_ms", "checksums"):
        assert key in d, key
    assert d["unit"] == "Gbp/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert d["higher_is_better"] is True and d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["parity"]["equal"] is True and d["parity"]["sample_genomes"] == 4 and d["parity"]["vs"] in ("reference", "port")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["unit"] == "Gbp/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_two_host_threads_two_streams_share_the_block_pool():
    """The caching allocator hands blocks from one stream's build to another's (two host threads, each with its own HIP
    stream, on one device): every hand-over is fenced on the device (StreamScope / events, api.hip), so every build must
    give the single-threaded result."""
    import threading

    import torch
    b1 = Batch.synthetic(24, 4, 40000, n_ancestors=3, snp_ppm=10000, seed=5)
    b2 = Batch.synthetic(16, 3, 70000, n_ancestors=2, snp_ppm=20000, seed=6)
    t1, t2 = np.arange(24) % 2 == 0, np.arange(16) % 3 == 0
    ref1, ref2 = b1.build_index(21, 200, t1).checksums(), b2.build_index(15, 50, t2).checksums()
    errors = []

    def work(batch, k, w, tar, ref, n):
        try:
            s = torch.cuda.Stream()
            for _ in range(n):
                ix = batch.build_index(k, w, tar, stream=int(s.cuda_stream))
                if ix.checksums() != ref:
                    errors.append((k, w))
                ix.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(b1, 21, 200, t1, ref1, 25)),
          threading.Thread(target=work, args=(b2, 15, 50, t2, ref2, 25)),
          threading.Thread(target=work, args=(b1, 21, 200, t1, ref1, 25))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def _resident_stats():
    import ctypes

    from seqwin_amd._lib import lib
    v = (ctypes.c_uint64 * 3)()
    lib.sw_resident_stats.restype = None
    lib.sw_resident_stats(v)
    return [int(x) for x in v]


def test_identity_test_of_the_resident_route_sees_permuted_node_ranges():
    """sw_get_penalty scores the index still resident in HBM only when the caller's arrays ARE the exported ones.  Until r05 the
    identity sums mixed the element index into kmers and nodes.hash only, so caller nodes whose start / stop had been permuted
    among nodes passed for the exported ones and were scored from HBM -- a silent divergence from filter.cpp:92-136, which
    walks kmers[start:stop) of the arrays it is GIVEN.  Now every term carries the index: such arrays are
    uploaded and give what the reference gives for them; the device checksums equal the numpy restatement on scored nodes
    (counts and penalty bits in the sum)."""
    paths = sorted((GOLDEN / "synth").glob("pan_*.fa"))
    tar = [i % 2 == 0 for i in range(len(paths))]
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    g = KmerGraph(paths, kmerlen=15, windowsize=20, n_cpu=2)
    hits = _resident_stats()[1]
    # two nodes with the same number of occurrences exchange their ranges: sizes, multisets of start and of stop unchanged
    size = (en["stop"] - en["start"]).astype(np.int64)
    i = int(np.flatnonzero(size == size[0])[1]) if np.count_nonzero(size == size[0]) > 1 else None
    assert i is not None
    nodes = g.nodes.copy()
    for f in ("start", "stop"):
        nodes[f][[0, i]] = g.nodes[f][[i, 0]]
    want = en.copy()
    for f in ("start", "stop"):
        want[f][[0, i]] = en[f][[i, 0]]
    oracle.get_penalty(ek, want, eo, tar)
    _get_penalty(g.kmers, nodes, g.record_offsets, tar)
    assert _resident_stats()[1] == hits, "permuted node ranges were taken for the exported arrays"
    assert np.array_equal(nodes, want)
    _get_penalty(g.kmers, g.nodes, g.record_offsets, tar)                    # the exported arrays themselves: resident
    assert _resident_stats()[1] == hits + 1
    b = Batch.from_fasta(paths, n_cpu=2)
    ix = b.build_index(15, 20, tar)
    K, N, E = ix.export()
    assert np.any(N["penalty"] != 0) and ix.checksums() == host_checksums(K, N, E)
    N2 = N.copy()
    N2["penalty"][[0, 1]] = N2["penalty"][[1, 0]]
    assert N["penalty"][0] == N["penalty"][1] or host_checksums(K, N2, E)[1] != ix.checksums()[1]


def test_resident_index_serves_get_penalty_and_filter_kmers(tmp_path, monkeypatch):
    """The index of the last build stays in HBM; get_penalty / filter_kmers use it instead of uploading the caller's arrays
    when -- and only when -- those still are the exported ones (host checksums); the size phase of filter_kmers keeps its
    device result for the data phase.  Results are the same on every route."""
    paths = sorted((GOLDEN / "synth").glob("pan_*.fa"))
    tar = [i % 2 == 0 for i in range(len(paths))]
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    scored = en.copy()
    oracle.get_penalty(ek, scored, eo, tar)
    g = KmerGraph(paths, kmerlen=15, windowsize=20, n_cpu=2)
    s0 = _resident_stats()
    assert s0[0] == len(g.kmers)
    nodes = g.nodes.copy()
    _get_penalty(g.kmers, nodes, g.record_offsets, tar)                     # a COPY of the exported nodes: same content -> resident
    assert np.array_equal(nodes, scored) and _resident_stats()[1] == s0[1] + 1
    _get_penalty(g.kmers, g.nodes, g.record_offsets, [not t for t in tar])  # again, other targets, already-scored nodes
    inv = en.copy()
    oracle.get_penalty(ek, inv, eo, [not t for t in tar])
    assert np.array_equal(g.nodes, inv) and _resident_stats()[1] == s0[1] + 2
    bad = g.kmers.copy()
    bad[5]["record_idx"] = 10_000                                            # not the exported array any more: uploaded, validated
    with pytest.raises(ValueError):
        _get_penalty(bad, g.nodes.copy(), g.record_offsets, tar)
    swapped = g.kmers.copy()
    swapped[[0, 1]] = swapped[[1, 0]]
    n2 = en.copy()
    try:
        _get_penalty(swapped, n2, g.record_offsets, tar)
        e2 = en.copy()
        oracle.get_penalty(swapped, e2, eo, tar)
        assert np.array_equal(n2, e2)
    except ValueError:
        pass                                                                 # (the swap may break the record order of a node)
    assert _resident_stats()[1] == s0[1] + 2
    used = frozenset(np.uint64(h) for h in en["hash"][::3])
    f1 = _filter_kmers(g.kmers, scored, used)
    f2 = oracle.filter_kmers(ek, scored, used)
    assert np.array_equal(f1[0], f2[0]) and np.array_equal(f1[1], f2[1])
    assert _resident_stats()[2] == s0[2] + 1                                 # one compute for the two phases, on resident kmers
    from seqwin_amd._lib import lib
    lib.sw_release_resident.restype = None
    lib.sw_release_resident()                                                # the HBM goes back at once; later calls upload
    assert _resident_stats()[0] == 0
    n3 = en.copy()
    _get_penalty(g.kmers, n3, g.record_offsets, tar)
    assert np.array_equal(n3, scored) and _resident_stats()[1] == s0[1] + 2
    monkeypatch.setenv("SEQWIN_AMD_NO_RESIDENT", "1")
    g2 = KmerGraph(paths, kmerlen=15, windowsize=20, n_cpu=2)
    assert _resident_stats()[0] == 0
    _get_penalty(g2.kmers, g2.nodes, g2.record_offsets, tar)
    assert np.array_equal(g2.nodes, scored)
    f3 = _filter_kmers(g2.kmers, scored, used)
    assert np.array_equal(f3[0], f2[0]) and np.array_equal(f3[1], f2[1])


@pytest.mark.parametrize("impl", ["own", "rocprim"])
def test_sort_pairs32_is_a_stable_radix_sort(impl, monkeypatch):
    """sw_sort_pairs32 = lsd_radix_sort (build_internals.cpp:76-110) as the node sort uses it: 32-bit keys with a 16-byte payload,
    csrc/radix.hip's pair passes (7168-element tiles, ranks from LDS atomics, look-back with epoch-tagged records -- several sorts
    in a row share the state buffer) or rocPRIM; sizes around the tile, long runs of equal keys, one key value, 8 ... 32 key bits."""
    import ctypes

    import torch
    from seqwin_amd._lib import c_u64, c_vp, check, lib
    monkeypatch.setenv("SEQWIN_AMD_SORT", impl)
    monkeypatch.setenv("SEQWIN_AMD_PAIR_SORT", impl)
    g = torch.Generator(device="cuda").manual_seed(5)
    for n, end_bit, hi in [(1, 32, 2**31), (5, 8, 2**31), (7167, 32, 2**31), (7168, 16, 2**31), (7169, 32, 2**31), (14_337, 24, 2**31),
                           (1_000_003, 32, 2**31), (3_000_000, 32, 1000), (2_000_000, 32, 1), (500_000, 16, 2**31)]:
        keys = (torch.randint(0, hi, (n,), dtype=torch.int64, device="cuda", generator=g) * (2_000_003 if hi <= 1000 else 1)).to(torch.int32)
        vals = torch.stack([torch.arange(n, device="cuda", dtype=torch.int32)] * 4, dim=1).contiguous()   # payload: where the pair stood
        vals[:, 1] ^= 0x5A5A5A5A
        vals[:, 3] = keys
        k, ka, v, va = keys.clone(), torch.empty_like(keys), vals.clone(), torch.empty_like(vals)
        flag = ctypes.c_int()
        check(lib.sw_sort_pairs32(c_vp(k.data_ptr()), c_vp(ka.data_ptr()), c_vp(v.data_ptr()), c_vp(va.data_ptr()), c_u64(n),
                                  c_u64(end_bit), c_vp(0), ctypes.byref(flag), None))
        ok, ov = (ka, va) if flag.value else (k, v)
        field = (keys.to(torch.int64) & 0xFFFFFFFF) & ((1 << end_bit) - 1)
        order = torch.sort(field, stable=True).indices
        assert torch.equal(ok, keys[order]) and torch.equal(ov, vals[order]), (impl, n, end_bit, hi)
    with pytest.raises(ValueError):
        check(lib.sw_sort_pairs32(c_vp(0), c_vp(0), c_vp(0), c_vp(0), c_u64(0), c_u64(12), c_vp(0), ctypes.byref(flag), None))


@pytest.mark.parametrize("impl", ["own", "rocprim"])
def test_sort_keys64_is_a_stable_radix_sort(impl, monkeypatch):
    """sw_sort_keys64 = lsd_radix_sort_key (build_internals.cpp:76-144) on the device: csrc/radix.hip (hand-written onesweep,
    16384-key tiles, ranks from LDS atomics -- SEQWIN_AMD_RADIX_RANK=ballot: from ballots --, decoupled look-back) or rocPRIM; any
    bit range, sizes around the tile, heavy ties."""
    import ctypes

    import torch
    from seqwin_amd._lib import c_u64, c_vp, check, lib
    monkeypatch.setenv("SEQWIN_AMD_SORT", impl)
    g = torch.Generator(device="cuda").manual_seed(3)
    for n, begin, end, hi in [(0, 0, 8, 2**20), (1, 0, 64, 2**62), (63, 5, 9, 2**12), (8191, 0, 16, 2**16), (8192, 3, 27, 2**30),
                              (8193, 0, 54, 2**54), (1_000_003, 46, 62, 2**62), (3_000_000, 0, 64, 2**62), (500_000, 0, 24, 4)]:
        keys = torch.randint(0, hi, (n,), dtype=torch.int64, device="cuda", generator=g)
        a, b = keys.clone(), torch.empty_like(keys)
        flag = ctypes.c_int()
        check(lib.sw_sort_keys64(c_vp(a.data_ptr()), c_vp(b.data_ptr()), c_u64(n), c_u64(begin), c_u64(end), c_vp(0), ctypes.byref(flag), None))
        out = b if flag.value else a
        field = (keys >> begin) & ((1 << (end - begin)) - 1) if end - begin < 64 else keys
        assert torch.equal(out, keys[torch.sort(field, stable=True).indices]), (n, begin, end)


_ORDER_GUARD_CHILD = r"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.environ["SW_ROOT"])
sys.path.insert(0, os.path.join(os.environ["SW_ROOT"], "tests"))
import oracle
from bench import SEED, write_fasta_sample
from seqwin_amd._lib import check, lib
from seqwin_amd.device import Batch

def trips():
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    check(lib.sw_order_guard_trips(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value

def mode():
    m = ctypes.c_int(-1)
    check(lib.sw_radix_rank_mode(ctypes.byref(m)))
    return m.value

ng, rpg, rl, k, w = 12, 5, 60000, 21, 200
b = Batch.synthetic(ng, rpg, rl, n_ancestors=3, snp_ppm=10000, seed=SEED)
tar = [i < ng // 2 for i in range(ng)]
paths, _ = write_fasta_sample(b, ng, os.environ["SW_TMP"])
ek, en, ee, eo, _ = oracle.build(paths, k, w)
oracle.get_penalty(ek, en, eo, tar)
out = {"mode_before": mode(), "trips_before": trips()}
os.environ["SEQWIN_AMD_FAULT_INJECT"] = "rank"          # from here on the LDS-atomic passes mis-rank neighbouring equal digits
for rnd in ("first", "second"):
    ix = b.build_index(k, w, tar)
    K, N, E = ix.export()
    out[rnd + "_equal"] = bool(np.array_equal(K, ek) and np.array_equal(N, en) and np.array_equal(E, ee))
    out[rnd + "_trips"] = trips()
    out[rnd + "_mode"] = mode()
    ix.close()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("which", ["nodes", "edges"])
def test_order_guard_detects_and_recovers_from_misranked_passes(which, tmp_path):
    """The radix passes rank by one LDS atomic per key and are stable only while the LDS unit serves the lanes of an atomic in lane
    order -- checked at start-up, but no architectural promise.  k_nodes and k_rle_keys therefore
    check the order of what they stream over in EVERY build: (hash, stream index) ascending, edge keys ascending -- the stability
    contract of lsd_radix_sort, build_internals.cpp:76-144.  SEQWIN_AMD_FAULT_INJECT=rank makes the first wave of every tile swap
    the ranks of neighbouring lanes with equal digits: the build must notice (sw_order_guard_trips), switch the device to ballot /
    rocPRIM ranking, sort again and still return the oracle's arrays; the next build stays on the safe ranking and trips nothing.
    which = nodes: the pair passes of the node sort are hit first (the edge sort then already runs on ballots);
    which = edges: the pairs go through rocPRIM, so the keys-only passes of the edge sort are the ones that mis-rank.
    (Own process: the demotion lasts for the rest of the process.)"""
    import json
    import os
    import subprocess
    import sys

    from conftest import ROOT
    env = dict(os.environ, SW_ROOT=str(ROOT), SW_TMP=str(tmp_path), SEQWIN_AMD_SORT="own", SEQWIN_AMD_UNSORT_DIRECT="4")
    env.pop("SEQWIN_AMD_RADIX_RANK", None)
    if which == "edges":
        env["SEQWIN_AMD_PAIR_SORT"] = "rocprim"
    r = subprocess.run([sys.executable, "-c", _ORDER_GUARD_CHILD], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    assert out["mode_before"] == 1 and out["trips_before"] == [0, 0], out        # a healthy device: atomics, nothing tripped
    assert out["first_equal"] and out["second_equal"], out                      # recovered: the oracle's arrays, twice
    hit = 0 if which == "nodes" else 1
    assert out["first_trips"][hit] == 1 and out["first_trips"][1 - hit] == 0, out
    assert out["first_mode"] == 0 and out["second_mode"] == 0, out              # demoted to ballots / rocPRIM
    assert out["second_trips"] == out["first_trips"], out                       # ... and the next build trips nothing
    assert "order guard" in r.stderr                                            # logged


def _device_gz_batches() -> int:
    import ctypes
    from seqwin_amd._lib import lib
    lib.sw_device_gz_batches.restype = ctypes.c_uint64
    return int(lib.sw_device_gz_batches())


_INGEST_ROUTE_CHILD = """
import sys, numpy as np
from seqwin_amd import KmerGraph
paths = open(sys.argv[1]).read().split()
g = KmerGraph(paths, kmerlen=17, windowsize=30, n_cpu=5)
np.savez(sys.argv[2], kmers=g.kmers, nodes=g.nodes, edges=g.edges, record_offsets=g.record_offsets)
"""


def test_streaming_ingest_buffer_routes_give_one_batch(tmp_path):
    """sw_build's streaming ingest (r05: the parsers pack into page-locked blocks that the DMA engine reads in place, and stay
    within a window of the assembly the sink thread is at): the default, every copy through the ring (pool of 0 MB), a pool
    that runs out after a few blocks with slabs smaller than the largest assembly's block, a window of one assembly, and files
    read 1 KiB at a time -- each in a process of its own (the pool lives as long as the process) -- give the oracle's graph."""
    import os
    import subprocess
    import sys

    from conftest import ROOT
    rng = random.Random(17)
    paths = []
    for a in range(48):
        txt = ""
        for r in range(rng.randrange(1, 4)):
            s = _randseq(rng, rng.choice([0, 40, 3000, 20000, 60000]))
            txt += f">r{r}_{a} d\n" + "\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + "\n"
        if a == 7:   # 1.25 MB of packed words: more than a 1 MiB slab
            big = np.random.default_rng(3).choice(np.frombuffer(b"ACGT", np.uint8), 5_000_000).tobytes().decode()
            txt += ">big\n" + "\n".join(big[i:i + 70] for i in range(0, len(big), 70)) + "\n"
        p = tmp_path / f"s{a}.fa"
        p.write_text(txt)
        paths.append(str(p))
    (tmp_path / "paths.txt").write_text("\n".join(paths))
    exp = oracle.build(paths, 17, 30)
    for i, extra in enumerate(({}, {"SEQWIN_AMD_PINNED_POOL_MB": "0"}, {"SEQWIN_AMD_PINNED_POOL_MB": "3", "SEQWIN_AMD_PINNED_SLAB_MB": "1"},
                               {"SEQWIN_AMD_INGEST_WINDOW": "1"}, {"SEQWIN_AMD_READ_BLOCK_KB": "1"})):
        out = tmp_path / f"g{i}.npz"
        env = dict(os.environ, PYTHONPATH=str(ROOT), **extra)
        r = subprocess.run([sys.executable, "-c", _INGEST_ROUTE_CHILD, str(tmp_path / "paths.txt"), str(out)], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        got = np.load(out)
        for name, want in zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4]):
            assert np.array_equal(got[name], want), (extra, name)


def test_device_gz_ingest_matches_host_route(tmp_path, monkeypatch):
    """.gz inputs inflated, parsed and packed ON THE DEVICE (csrc/ingest_dev.hip: the gzip branch of fasta_reader.cpp:109-203
    and the parse of :41-95, one file per lane) give the batch the host route gives: records, ids, graph -- and the oracle's
    graph.  Stored / fixed / dynamic blocks (compresslevel 0 / 1 / 9 and a hand-made fixed-code stream), header fields (FNAME,
    FEXTRA, FCOMMENT, FHCRC), CRLF, blank and whitespace-only lines, lower case, IUPAC, N runs, an id that ends the file,
    records without sequence, long matches (repeats) and incompressible text."""
    import zlib
    rng = random.Random(5)
    texts = []
    for a in range(70):                      # more files than one wave has lanes
        txt = "" if a % 9 else "\n  \n"
        for r in range(rng.randrange(0, 5)):
            n = rng.choice([0, 1, 31, 32, 33, 64, 100, 1000, 5000, 40000, 130000])
            s = _randseq(rng, n)
            if a % 7 == 3 and n > 1000:
                s = (s[:500] * (n // 500 + 1))[:n]                  # repeats: matches of the maximum length, distances up to 500
            txt += f">rec{r}_{a}" + rng.choice([" some description", "\tx", "", " "]) + rng.choice(["\n", "\r\n"])
            width = rng.choice([60, 70, 80, 7, 1000, 10**6])
            for i in range(0, len(s), width):
                txt += s[i:i + width] + rng.choice(["\n", "\r\n", " \n", "\n\n"])
        if a % 11 == 5:
            txt += ">last_id_without_newline"
        if a % 13 == 6:
            txt = txt.rstrip("\n")
        texts.append(txt.encode())
    texts.append(b"")                                                # an empty file
    texts.append(b">only_header\n")
    texts.append(bytes(rng.getrandbits(8) | 0x40 for _ in range(50000)).replace(b">", b"A"))   # no header: ...
    texts[-1] = b">noise\n" + texts[-1]                              # ... under one (incompressible: stored blocks at level 9 too)
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / f"a{i}.fa.gz"
        level = [0, 1, 6, 9][i % 4]
        if i % 5 == 0:                                               # hand-made member: FEXTRA + FNAME + FCOMMENT + FHCRC, fixed-code stream
            co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_FIXED if i % 10 == 0 else zlib.Z_DEFAULT_STRATEGY)
            body = co.compress(t) + co.flush()
            head = bytes([0x1F, 0x8B, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 255]) + (5).to_bytes(2, "little") + b"extra" + b"name.fa\0" + b"a comment\0"
            head += (zlib.crc32(head) & 0xFFFF).to_bytes(2, "little")
            p.write_bytes(head + body + zlib.crc32(t).to_bytes(4, "little") + (len(t) & 0xFFFFFFFF).to_bytes(4, "little"))
        else:
            with gzip.GzipFile(p, "wb", compresslevel=level) as f:
                f.write(t)
        paths.append(p)
    k, w = 15, 20
    monkeypatch.setenv("SEQWIN_AMD_DEVICE_INFLATE", "0")
    host = KmerGraph(paths, kmerlen=k, windowsize=w, n_cpu=3)
    monkeypatch.setenv("SEQWIN_AMD_DEVICE_INFLATE", "1")
    n0 = _device_gz_batches()
    dev = KmerGraph(paths, kmerlen=k, windowsize=w, n_cpu=3)
    assert _device_gz_batches() == n0 + 1                            # the device route was taken, not declined
    assert np.array_equal(dev.record_offsets, host.record_offsets) and [list(x) for x in dev.record_ids] == [list(x) for x in host.record_ids]
    assert np.array_equal(dev.kmers, host.kmers) and np.array_equal(dev.nodes, host.nodes) and np.array_equal(dev.edges, host.edges)
    exp = oracle.build(paths, k, w)
    assert_graph_equal((dev.kmers, dev.nodes, dev.edges, dev.record_offsets, dev.record_ids),
                       dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])
    b_dev = Batch.from_fasta(paths, n_cpu=2)
    assert _device_gz_batches() == n0 + 2
    monkeypatch.setenv("SEQWIN_AMD_DEVICE_INFLATE", "0")
    b_host = Batch.from_fasta(paths, n_cpu=2)
    i_dev, i_host = b_dev.info(), b_host.info()
    assert all(i_dev[key] == i_host[key] for key in ("n_assemblies", "n_records", "total_bp"))   # (device_bytes: the host route over-allocates)
    ix_dev, ix_host = b_dev.build_index(k, w), b_host.build_index(k, w)
    assert ix_dev.checksums() == ix_host.checksums()

    # what the device route declines goes through the host route, with the host route's outcome
    monkeypatch.setenv("SEQWIN_AMD_DEVICE_INFLATE", "1")
    two = tmp_path / "two_members.fa.gz"                              # concatenated members: gzread reads them as one file
    two.write_bytes(gzip.compress(b">m1\nACGTACGTACGTACGTACGTAAAC\n") + gzip.compress(b">m2\nTTGACCAGTACGGGATACCAGT\n"))
    g = KmerGraph([paths[1], two], kmerlen=5, windowsize=3, n_cpu=1)
    assert _device_gz_batches() == n0 + 2 and [len(x) for x in g.record_ids][1] == 2
    exp = oracle.build([paths[1], two], 5, 3)
    assert np.array_equal(g.kmers, exp[0])
    bad = tmp_path / "bad_crc.fa.gz"
    raw = bytearray(gzip.compress(b">c\n" + b"ACGTTGCA" * 400 + b"\n", compresslevel=0))
    raw[40] ^= 0x02                                                   # a stored byte changed: same length, CRC-32 differs
    bad.write_bytes(bytes(raw))
    with pytest.raises(RuntimeError, match="gzip read error"):
        KmerGraph([paths[1], bad], kmerlen=5, windowsize=3, n_cpu=1)
    ctl = tmp_path / "ctl.fa.gz"
    ctl.write_bytes(gzip.compress(b">c\nACGT\x03ACGT\n"))
    with pytest.raises(ValueError, match="control byte"):
        KmerGraph([ctl], kmerlen=3, windowsize=1, n_cpu=1)
    assert _device_gz_batches() == n0 + 2


# ---- one sw_build over several devices (SEQWIN_DEVICES, csrc/multi.hip) -------------------------------------------------------

@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,0,0,0,0,0,0,0"])
@pytest.mark.parametrize("route", [None, "requests"])
def test_multi_device_build_equals_single_device(tmp_path, smoke_paths, monkeypatch, devices, route):
    """KmerGraph(paths, ...) under SEQWIN_DEVICES -- one host thread and one stream per listed device inside the ONE sw_build
    call, the reference's partition of the assemblies over its workers (build.cpp:342-367), tuple / rank / key exchanges by
    peer copies -- must give the arrays of the single-device build (and of the oracle), whatever the number of shards: the
    reference's thread-count invariance (tests/smoke/test_graph.py:67-127).  One GPU here, so the listed devices are logical
    shards on card 0; more shards than assemblies are clamped (eight devices, four or six assemblies)."""
    if route:
        monkeypatch.setenv("SEQWIN_DIST_HASH_ROUTE", route)
    synth = sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))
    lc = tmp_path / "lc.fa"     # tandem repeats: pairs that repeat inside an assembly (the candidate rows), every window a tie
    lc.write_text(">at\n" + "AT" * 4000 + "\n>rep7\n" + "ACGGTCA" * 2000 + "\n>mix\n" + "A" * 500 + "N" + "C" * 700 + "ACGT" * 300 + "\n")
    for paths, k, w in ((smoke_paths, 17, 10), (smoke_paths, 21, 200), (synth, 15, 20), (synth + [lc, lc], 11, 5), ([lc], 7, 3)):
        monkeypatch.delenv("SEQWIN_DEVICES", raising=False)
        one = _build(paths, k, w, n_cpu=2)
        monkeypatch.setenv("SEQWIN_DEVICES", devices)
        many = _build(paths, k, w, n_cpu=3)
        for a, b in zip(one[:4], many[:4]):
            assert a.dtype == b.dtype and np.array_equal(a, b), (devices, route, k, w)
        assert one[4] == many[4]
        exp = oracle.build(paths, k, w)
        assert_graph_equal(many, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])
        if len(many[1]) and len(paths) >= 2:
            # the call Seqwin makes next: the slices are still on their devices and are scored there (no upload) ...
            tar = [i % 2 == 0 for i in range(len(paths))]
            oracle.get_penalty(exp[0], exp[1], exp[3], tar)
            hits = _resident_stats()[1]
            _get_penalty(many[0], many[1], many[3], tar)
            assert np.array_equal(many[1], exp[1])
            assert _resident_stats()[1] == hits + 1, "the resident slices did not serve get_penalty"
            # ... again with other targets (the resident nodes now hold counts; the identity covers hash / start / stop) ...
            tar2 = [i % 3 == 0 for i in range(len(paths))]
            if any(tar2) and not all(tar2):
                exp2 = [a.copy() for a in exp[:2]]
                oracle.get_penalty(exp2[0], exp2[1], exp[3], tar2)
                _get_penalty(many[0], many[1], many[3], tar2)
                assert np.array_equal(many[1], exp2[1]) and _resident_stats()[1] == hits + 2
            # ... and once the slices are gone: the ordinary upload route, same result
            from seqwin_amd._lib import lib as _l
            _l.sw_release_resident.restype = None
            _l.sw_release_resident()
            n2 = many[1].copy()
            n2["n_tar"] = 0
            n2["n_neg"] = 0
            n2["penalty"] = 0.0
            _get_penalty(many[0], n2, many[3], tar)
            assert np.array_equal(n2, exp[1]) and _resident_stats()[1] == hits + (2 if any(tar2) and not all(tar2) else 1)


def test_multi_device_build_on_a_synthetic_job(tmp_path, monkeypatch):
    """The same on 24 genomes x 4 contigs x 60 kbp (5.8 Mbp, 58 k minimizers; FASTA written from the device generator): three and
    five logical devices against the single-device build."""
    from test_gpu_fullsize import write_fasta_sample
    b = Batch.synthetic(24, 4, 60_000, n_ancestors=3, snp_ppm=20_000, seed=11)
    paths, _ = write_fasta_sample(b, 24, str(tmp_path))
    b.close()
    one = _build(paths, 21, 200, n_cpu=4)
    for devices in ("0,0,0", "0,0,0,0,0"):
        monkeypatch.setenv("SEQWIN_DEVICES", devices)
        many = _build(paths, 21, 200, n_cpu=4)
        monkeypatch.delenv("SEQWIN_DEVICES")
        assert all(np.array_equal(a, c) for a, c in zip(one[:4], many[:4])) and one[4] == many[4], devices
    # r05: the exchanges without peer access -- every pull staged through the pulling worker's pinned host buffer (the route a pair
    # of GPUs without hipDeviceCanAccessPeer takes; forced here, on logical shards of one card) -- and the route in the log
    import logging
    records = []

    class _Grab(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())
    h = _Grab(level=logging.INFO)
    root = logging.getLogger()
    old_level = root.level
    root.addHandler(h)
    root.setLevel(logging.INFO)
    try:
        for no_p2p, expect in (("1", "staged through pinned host memory (SEQWIN_MULTI_NO_P2P=1)"), ("0", "logical shards of one device")):
            monkeypatch.setenv("SEQWIN_DEVICES", "0,0,0,0")
            monkeypatch.setenv("SEQWIN_MULTI_NO_P2P", no_p2p)
            del records[:]
            many = _build(paths, 21, 200, n_cpu=4)
            assert all(np.array_equal(a, c) for a, c in zip(one[:4], many[:4])) and one[4] == many[4], no_p2p
            assert any("multi-device build: 4 workers on devices [0,0,0,0]" in m and expect in m for m in records), records
    finally:
        root.removeHandler(h)
        root.setLevel(old_level)


def test_build_splits_itself_when_the_occurrences_outgrow_32_bit_indices(tmp_path, monkeypatch):
    """The reference indexes occurrences with size_t (cpp/include/seqwin/graph.hpp:28-41); here a device addresses 2^32 - 2 of
    them.  Until r05 a larger job was a RuntimeError; now sw_build splits it into logical shards by itself
    -- the SEQWIN_DEVICES machinery on the current card, or the listed devices taken round robin -- until every shard and slice
    fits, and says so in the log.  SEQWIN_AMD_OCC_CAP lowers the bound so that a 58 k-minimizer job takes that route: plain build
    (1 -> several shards), a build that was already sharded (SEQWIN_DEVICES=0,0 -> more), low_memory on top; the arrays are the
    single-device build's.  A job that cannot be split further (one assembly) still fails, with the reference-free message."""
    import logging
    from test_gpu_fullsize import write_fasta_sample
    b = Batch.synthetic(24, 4, 60_000, n_ancestors=3, snp_ppm=20_000, seed=11)
    paths, _ = write_fasta_sample(b, 24, str(tmp_path))
    b.close()
    one = _build(paths, 21, 200, n_cpu=4)
    n_occ = len(one[0])
    records = []

    class _Grab(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())
    h = _Grab(level=logging.INFO)
    root = logging.getLogger()
    old_level = root.level
    root.addHandler(h)
    root.setLevel(logging.INFO)
    try:
        for cap, devices, low in ((n_occ // 3, None, False), (n_occ // 5, "0,0", False), (n_occ // 3, None, True), (n_occ // 2 + 1, "0,0,0", True)):
            monkeypatch.setenv("SEQWIN_AMD_OCC_CAP", str(cap))
            if devices:
                monkeypatch.setenv("SEQWIN_DEVICES", devices)
            else:
                monkeypatch.delenv("SEQWIN_DEVICES", raising=False)
            if low:
                monkeypatch.setenv("SEQWIN_AMD_LOWMEM_CHUNK_MBP", "1")
            del records[:]
            many = _build(paths, 21, 200, n_cpu=4, low_memory=low)
            assert all(np.array_equal(a, c) for a, c in zip(one[:4], many[:4])) and one[4] == many[4], (cap, devices, low)
            if cap < n_occ // 2 or not devices:
                assert any("splitting the job into" in m for m in records), records
            assert any("split automatically" in m or "SEQWIN_DEVICES" in m for m in records), records
            if low:
                assert any("streams its shard through HBM in chunks" in m for m in records), records
            monkeypatch.delenv("SEQWIN_AMD_LOWMEM_CHUNK_MBP", raising=False)
        monkeypatch.delenv("SEQWIN_DEVICES", raising=False)
        monkeypatch.setenv("SEQWIN_AMD_OCC_CAP", "100")
        with pytest.raises(RuntimeError, match="SEQWIN_AMD_OCC_CAP minimizer occurrences on one device"):
            KmerGraph(paths[:1], kmerlen=21, windowsize=200, n_cpu=1)
    finally:
        root.removeHandler(h)
        root.setLevel(old_level)


@pytest.mark.parametrize("chunk_mbp", ["0", "1", "4096"])
def test_pipelined_build_matches_standard(tmp_path, monkeypatch, chunk_mbp):
    """SEQWIN_AMD_PIPELINE=1: the host threads parse the files in consecutive chunks while a second host thread drives the device
    through the previous chunk (plan, sketch, ordered tuples); the index is built from the concatenated tuple stream.  The
    reference interleaves reading and minimizing per assembly (build.cpp:98-256); the result must not depend on it: one assembly
    per chunk, a few per chunk, everything in one chunk -- gz files, empty records, low complexity -- against the oracle."""
    from test_gpu_fullsize import write_fasta_sample
    monkeypatch.setenv("SEQWIN_AMD_PIPELINE", "1")
    monkeypatch.setenv("SEQWIN_AMD_PIPELINE_CHUNK_MBP", chunk_mbp)
    synth = sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))
    lc = tmp_path / "lc.fa"
    lc.write_text(">at\n" + "AT" * 4000 + "\n>rep7\n" + "ACGGTCA" * 2000 + "\n>mix\n" + "A" * 500 + "N" + "C" * 700 + "ACGT" * 300 + "\n")
    for paths, k, w in ((synth, 15, 20), (synth + [lc, lc], 11, 5)):
        exp = oracle.build(paths, k, w)
        many = _build(paths, k, w, n_cpu=3)
        assert_graph_equal(many, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])
    b = Batch.synthetic(24, 4, 60_000, n_ancestors=3, snp_ppm=20_000, seed=11)
    paths, _ = write_fasta_sample(b, 24, str(tmp_path))
    b.close()
    got = _build(paths, 21, 200, n_cpu=4)
    monkeypatch.delenv("SEQWIN_AMD_PIPELINE")
    one = _build(paths, 21, 200, n_cpu=4)
    assert all(np.array_equal(a, c) for a, c in zip(one[:4], got[:4])) and one[4] == got[4]
    with pytest.raises(RuntimeError, match="Unable to open FASTA"):
        monkeypatch.setenv("SEQWIN_AMD_PIPELINE", "1")
        _build(paths[:6] + [tmp_path / "missing.fa"] + paths[6:], 21, 200, n_cpu=2)


def test_pool_debug_mode_detects_a_block_written_after_its_release(tmp_path):
    """SEQWIN_AMD_POOL_DEBUG=1 (the soak mode of the device pool: releases wait for the device and poison, reuses wait and check) must
    DETECT -- a soak that reports "0 blocks written after their release" only means something if a block that IS written after its
    release is reported.  The test hook SEQWIN_AMD_FAULT_INJECT=pool writes one word into every 64th released block behind the
    poison, as a kernel still queued at the release would: builds then fail with the offset of the damage and sw_pool_debug_stats
    counts the blocks; without the injection the same builds under the debug mode give the oracle's arrays."""
    import subprocess
    import sys
    code = (
        "import ctypes, sys, numpy as np\n"
        "from seqwin_amd import KmerGraph\n"
        "from seqwin_amd._lib import lib\n"
        "paths = sys.argv[1:]\n"
        "errors, g = 0, None\n"
        "for rep in range(12):\n"
        "    try:\n"
        "        g = KmerGraph(paths, kmerlen=15, windowsize=20, n_cpu=2)\n"
        "    except RuntimeError as e:\n"
        "        assert 'POOL_DEBUG' in str(e) and 'offset 1024' in str(e), e\n"
        "        errors += 1\n"
        "st = (ctypes.c_uint64 * 3)()\n"
        "lib.sw_pool_debug_stats(st)\n"
        "print('debug', st[0], 'violations', st[1], 'errors', errors, 'kmers', len(g.kmers) if g is not None else -1)\n")
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa"))]
    env = dict(os.environ, SEQWIN_AMD_POOL_DEBUG="1")
    r = subprocess.run([sys.executable, "-c", code] + paths, capture_output=True, text=True, env=dict(env, SEQWIN_AMD_FAULT_INJECT="pool"), timeout=300,
                       cwd=str(Path(__file__).resolve().parent.parent))
    assert r.returncode == 0, r.stderr[-2000:]
    f = r.stdout.split()
    assert f[1] == "1" and int(f[3]) >= 1 and int(f[5]) >= 1, r.stdout
    assert "was written AFTER its release" in r.stderr
    r = subprocess.run([sys.executable, "-c", code] + paths, capture_output=True, text=True, env=env, timeout=300, cwd=str(Path(__file__).resolve().parent.parent))
    assert r.returncode == 0, r.stderr[-2000:]
    f = r.stdout.split()
    ek = oracle.build(paths, 15, 20)[0]
    assert f[1] == "1" and f[3] == "0" and f[5] == "0" and int(f[7]) == len(ek), r.stdout


def test_low_memory_is_honoured_under_seqwin_devices(tmp_path, monkeypatch):
    """low_memory / SEQWIN_AMD_HBM_BUDGET_GB under SEQWIN_DEVICES: every worker streams ITS shard through HBM in chunks
    (sw_occ_sketch_paths; until r05 the request was answered with a warning) -- chunk boundaries inside a shard, gz files, shards of
    one assembly -- and the arrays are the standard build's and the oracle's (build.cpp:264-325: the low-memory second pass changes
    the peak, never the result)."""
    synth = sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))
    exp = oracle.build(synth, 15, 20)
    for devices, chunk in (("0,0", "0"), ("0,0,0", "1"), ("0,0,0,0,0,0,0,0", "0")):
        monkeypatch.setenv("SEQWIN_DEVICES", devices)
        monkeypatch.setenv("SEQWIN_AMD_LOWMEM_CHUNK_MBP", chunk)     # 0: one assembly per chunk
        many = _build(synth, 15, 20, n_cpu=3, low_memory=True)
        assert_graph_equal(many, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])


def test_multi_device_build_without_any_record_or_minimizer(tmp_path, monkeypatch):
    """Assemblies without a record, or without a k-mer: every slice of the multi-device build is empty (found by the SEQWIN_DEVICES
    fuzz campaign: a slice that received nothing reports no rank marks, which must not be taken for "2^31 nodes")."""
    empty = [tmp_path / f"e{i}.fa" for i in range(3)]
    for p in empty:
        p.write_text("")
    short = [tmp_path / f"s{i}.fa" for i in range(3)]
    for i, p in enumerate(short):
        p.write_text(f">r{i}\nACGTACGT\n>q{i}\nNNNN\n")
    monkeypatch.setenv("SEQWIN_DEVICES", "0,0,0")
    for paths in (empty, short, empty + short):
        kmers, nodes, edges, offs, ids = _build(paths, 21, 200, n_cpu=2)
        assert len(kmers) == 0 and len(nodes) == 0 and len(edges) == 0
        exp = oracle.build(paths, 21, 200)
        assert np.array_equal(offs, exp[3]) and [tuple(t) for t in ids] == [tuple(t) for t in exp[4]]
    one = tmp_path / "one.fa"
    one.write_text(">a\n" + "ACGGTCA" * 200 + "\n")                       # all minimizers come from one of the shards
    got = _build(empty + [one] + short, 7, 5, n_cpu=2)
    exp = oracle.build(empty + [one] + short, 7, 5)
    assert_graph_equal(got, dict(zip(("kmers", "nodes", "edges", "record_offsets"), exp[:4])), [list(t) for t in exp[4]])


def test_seqwin_devices_is_validated(smoke_paths, monkeypatch):
    for bad in ("0,99", "zero", "0;1", "-1,0"):
        monkeypatch.setenv("SEQWIN_DEVICES", bad)
        with pytest.raises(ValueError, match="SEQWIN_DEVICES"):
            _build(smoke_paths, 17, 10, n_cpu=1)
    monkeypatch.setenv("SEQWIN_DEVICES", "0")          # one device: the ordinary build
    assert len(_build(smoke_paths, 17, 10, n_cpu=1)[0])


@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_pipelined_download_equals_plain_copy(tmp_path, smoke_paths, monkeypatch, devices):
    """sw_graph_export / sw_index_export bring large results to the host through a ring of pinned slots and several copying
    threads (api.hip: download); below 256 MiB they use one hipMemcpy per array.  Both must give the same bytes: the ring is
    forced on small graphs here (every chunk partial, arrays of a few bytes, empty arrays), with and without slices on several
    logical devices, and runs at its own size on a 2 Gbp batch (chunks of 8 MiB, whole and partial)."""
    if devices:
        monkeypatch.setenv("SEQWIN_DEVICES", devices)
    synth = sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))
    empty = tmp_path / "empty.fa"
    empty.write_text("")
    for paths, k, w in ((smoke_paths, 17, 10), (synth, 15, 20), ([empty, empty], 21, 200)):
        monkeypatch.setenv("SEQWIN_AMD_PLAIN_DOWNLOAD", "1")
        plain = _build(paths, k, w, n_cpu=2)
        monkeypatch.delenv("SEQWIN_AMD_PLAIN_DOWNLOAD")
        monkeypatch.setenv("SEQWIN_AMD_DOWNLOAD_PIPELINE_MB", "0")
        ring = _build(paths, k, w, n_cpu=2)
        monkeypatch.setenv("SEQWIN_AMD_DOWNLOAD_SLOT_KB", "1")      # many chunks: 85 packed nodes / 51 packed edges / 128 kmers each
        small_slots = _build(paths, k, w, n_cpu=2)
        monkeypatch.setenv("SEQWIN_AMD_EXPORT_WHOLE", "1")          # ... and nodes / edges as they are (until r05a) instead of packed
        whole = _build(paths, k, w, n_cpu=2)
        monkeypatch.delenv("SEQWIN_AMD_EXPORT_WHOLE")
        monkeypatch.delenv("SEQWIN_AMD_DOWNLOAD_SLOT_KB")
        monkeypatch.delenv("SEQWIN_AMD_DOWNLOAD_PIPELINE_MB")
        for other in (ring, small_slots, whole):
            for a, b in zip(plain[:4], other[:4]):
                assert a.dtype == b.dtype and np.array_equal(a, b), (k, w)
        # sw_get_penalty's nodes come back by the same routes (resident index or uploaded arrays): one result, and the oracle's
        if len(plain[1]) and len(paths) >= 2:
            tar = np.array([i % 2 == 0 for i in range(len(paths))], np.bool_)
            exp = oracle.build(paths, k, w)
            oracle.get_penalty(exp[0], exp[1], exp[3], list(tar))
            for env in ({}, {"SEQWIN_AMD_DOWNLOAD_PIPELINE_MB": "0"}, {"SEQWIN_AMD_DOWNLOAD_PIPELINE_MB": "0", "SEQWIN_AMD_DOWNLOAD_SLOT_KB": "1"},
                        {"SEQWIN_AMD_NO_RESIDENT": "1"}):
                for key, val in env.items():
                    monkeypatch.setenv(key, val)
                g = _build(paths, k, w, n_cpu=2)
                _get_penalty(g[0], g[1], g[3], tar, 2)
                for key in env:
                    monkeypatch.delenv(key)
                assert np.array_equal(g[1], exp[1]), (k, w, env)
    if devices:
        return
    from seqwin_amd.device import Batch
    b = Batch.synthetic(400, 50, 100000, n_ancestors=7, snp_ppm=10000, seed=11)
    ix = b.build_index(21, 200, [i % 2 == 0 for i in range(400)])
    monkeypatch.setenv("SEQWIN_AMD_PLAIN_DOWNLOAD", "1")
    plain = ix.export()
    monkeypatch.delenv("SEQWIN_AMD_PLAIN_DOWNLOAD")
    ring = ix.export()
    assert sum(a.nbytes for a in ring) > 256 << 20
    for a, b2 in zip(plain, ring):
        assert np.array_equal(a, b2)
