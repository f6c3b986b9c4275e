"""GPU (-m gpu): the device side of the multi-GPU merge (sw_index_occ_rows / _edge_rows / _splits /
sw_index_merge) on ONE GPU: P shard batches are built one after another, their rows are routed by
hand exactly as all_to_all_single would route them, every owner's slice is merged, and the
concatenation must equal the single-batch index bit for bit (shard-count invariance)."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN
from seqwin_amd import dist as swdist
from seqwin_amd.device import Batch

pytestmark = pytest.mark.gpu


def _route_and_merge(paths, world, k, w, tar):
    eng = swdist.HipEngine()
    parts = swdist.partition_assemblies(len(paths), world)
    shards = [Batch.from_fasta(paths[a:b], n_cpu=2) for a, b in parts]
    local = [s.build_index(k, w, None) for s in shards]
    offs = [s.records()[0] for s in shards]
    rec_base, glob, total = [], [np.zeros(1, np.uint32)], 0
    for o in offs:
        rec_base.append(total)
        glob.append(o[1:] + np.uint32(total))
        total += int(o[-1])
    record_offsets = np.concatenate(glob)
    nb, eb = swdist.hash_bounds(world)
    occ, edg, ocut, ecut = [], [], [], []
    for r, ix in enumerate(local):
        occ.append(eng.occ_rows(ix, rec_base[r]))
        edg.append(eng.edge_rows(ix))
        osp, esp = eng.splits(ix, nb, eb)
        nk, _, ne = ix.sizes()
        ocut.append([0] + osp + [nk])
        ecut.append([0] + esp + [ne])
    kmers, nodes, edges, base = [], [], [], 0
    for owner in range(world):
        r_occ = torch.cat([occ[r][ocut[r][owner]:ocut[r][owner + 1]] for r in range(world)])
        r_edg = torch.cat([edg[r][ecut[r][owner]:ecut[r][owner + 1]] for r in range(world)])
        m = eng.merge(r_occ, r_edg, base, record_offsets, tar)
        K, N, E = m.export()
        kmers.append(K); nodes.append(N); edges.append(E)
        base += len(K)
    return np.concatenate(kmers), np.concatenate(nodes), np.concatenate(edges), record_offsets


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_routed_merge_equals_single_batch(world):
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    tar = [i % 3 != 0 for i in range(len(paths))]
    for k, w in [(15, 20), (21, 200), (11, 5)]:
        got = _route_and_merge(paths, world, k, w, tar)
        ek, en, ee, eo, _ = oracle.build(paths, k, w)
        oracle.get_penalty(ek, en, eo, tar)
        assert np.array_equal(got[0], ek) and np.array_equal(got[1], en), (world, k, w)
        assert np.array_equal(got[2], ee) and np.array_equal(got[3], eo), (world, k, w)


def test_world1_sharded_index_matches_direct():
    b = Batch.synthetic(24, 4, 40000, n_ancestors=3, snp_ppm=10000, seed=5)
    tar = np.arange(24) % 2 == 0
    direct = b.build_index(21, 200, tar)
    for build in (swdist.build_sharded_index, swdist.build_sharded_index_merge):
        sharded = build(swdist.Shard(b, 0, 24), 21, 200, tar)
        for x, y in zip(direct.export(), sharded.export()):
            assert np.array_equal(x, y)
        assert sharded.timings()["n_occ_local"] == direct.sizes()[0]
    os.environ["SEQWIN_DIST_HASH_ROUTE"] = "requests"
    try:
        sharded = swdist.build_sharded_index(swdist.Shard(b, 0, 24), 21, 200, tar)
    finally:
        del os.environ["SEQWIN_DIST_HASH_ROUTE"]
    for x, y in zip(direct.export(), sharded.export()):
        assert np.array_equal(x, y)


def _gloo_worker(rank, world, port, paths, k, w, tar, out_path, mode):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seqwin_amd.device import set_device
        set_device(0)
        start, end = swdist.partition_assemblies(len(paths), world)[rank]
        shard = swdist.Shard(Batch.from_fasta(paths[start:end], n_cpu=2), start, len(paths))
        if mode == "tuples_requests":
            os.environ["SEQWIN_DIST_HASH_ROUTE"] = "requests"
        build = swdist.build_sharded_index if mode.startswith("tuples") else swdist.build_sharded_index_merge
        sharded = build(shard, k, w, tar, engine=swdist.HipEngine("host"))
        full = sharded.gather(0)
        if rank == 0:
            np.savez(out_path, kmers=full[0], nodes=full[1], edges=full[2], record_offsets=full[3])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tuples", "tuples_requests", "merge"])
@pytest.mark.parametrize("world", [2, 3])
def test_two_processes_one_gpu_real_collectives(tmp_path, world, mode):
    """The full multi-process path -- HIP engine in every process, real all_to_all_single / all_gather
    (over gloo with host staging, since this box has one GPU) -- must reproduce the single-process result."""
    import socket

    import torch.multiprocessing as mp
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    tar = [i % 2 == 0 for i in range(len(paths))]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "merged.npz"
    mp.spawn(_gloo_worker, nprocs=world, args=(world, port, paths, 15, 20, tar, str(out), mode), join=True)
    got = np.load(out)
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(got["kmers"], ek) and np.array_equal(got["nodes"], en)
    assert np.array_equal(got["edges"], ee) and np.array_equal(got["record_offsets"], eo)


@pytest.mark.parametrize("mode", ["tuples", "tuples_requests"])
def test_four_processes_with_an_empty_shard(tmp_path, mode):
    """Four ranks on three assemblies: the last rank holds no assembly (an empty batch, empty tuple stream, nothing to send)
    while it still owns a hash range and an edge range -- HIP engine in every process, real collectives over gloo."""
    import socket

    import torch.multiprocessing as mp
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa"))[:3]]
    tar = [True, False, True]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "merged.npz"
    mp.spawn(_gloo_worker, nprocs=4, args=(4, port, paths, 15, 20, tar, str(out), mode), join=True)
    got = np.load(out)
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(got["kmers"], ek) and np.array_equal(got["nodes"], en)
    assert np.array_equal(got["edges"], ee) and np.array_equal(got["record_offsets"], eo)


def _rccl_worker(rank, port, paths, k, w, tar, out_path):
    import os
    os.environ["SEQWIN_DIST_FORCE_COLLECTIVES"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import importlib

    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import seqwin_amd.dist as d
        importlib.reload(d)
        from seqwin_amd.device import set_device
        set_device(0)
        shard = d.Shard(Batch.from_fasta(paths, n_cpu=2), 0, len(paths))
        sharded = d.build_sharded_index(shard, k, w, tar)          # every collective is issued, over RCCL
        kmers, nodes, edges = sharded.export()
        ch, cnt, cnn, cpen = d.count_nodes_allreduce(shard, k, w, tar)   # the count-only path: all-gather + ONE all_reduce, over RCCL
        np.savez(out_path, kmers=kmers, nodes=nodes, edges=edges, record_offsets=sharded.record_offsets, c_hash=ch, c_tar=cnt,
                 c_neg=cnn, c_pen=cpen)
    finally:
        dist.destroy_process_group()


def test_count_only_allreduce_on_the_gpu(tmp_path):
    """dist.count_nodes_allreduce with the HIP engine (device-resident occurrence rows, torch ops on the GPU for the counts, the f64
    penalty computed there): single process, and two / three processes sharing the GPU with real collectives over gloo -- against
    the nodes get_penalty leaves (bit-identical, penalty included)."""
    import socket

    import torch.multiprocessing as mp
    b = Batch.synthetic(24, 4, 40000, n_ancestors=3, snp_ppm=10000, seed=5)
    tar = np.arange(24) < 9
    direct = b.build_index(21, 200, tar)
    N = direct.export()[1]
    h, nt, nn, pen = swdist.count_nodes_allreduce(swdist.Shard(b, 0, 24), 21, 200, tar)
    assert np.array_equal(h, N["hash"]) and np.array_equal(nt, N["n_tar"]) and np.array_equal(nn, N["n_neg"])
    assert np.array_equal(pen, N["penalty"])
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    tar2 = [i < 3 for i in range(len(paths))]
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    oracle.get_penalty(ek, en, eo, tar2)
    for world in (2, 3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        out = str(tmp_path / f"c{world}")
        mp.spawn(_gloo_counts_worker, nprocs=world, args=(world, port, paths, 15, 20, tar2, out), join=True)
        for r in range(world):
            got = np.load(out + f".r{r}.npz")
            assert np.array_equal(got["hash"], en["hash"]) and np.array_equal(got["n_tar"], en["n_tar"])
            assert np.array_equal(got["n_neg"], en["n_neg"]) and np.array_equal(got["penalty"], en["penalty"])


def _gloo_counts_worker(rank, world, port, paths, k, w, tar, out_path):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SEQWIN_DIST_SELFCHECK_MB"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seqwin_amd.device import set_device
        set_device(0)
        start, end = swdist.partition_assemblies(len(paths), world)[rank]
        shard = swdist.Shard(Batch.from_fasta(paths[start:end], n_cpu=2), start, len(paths))
        h, nt, nn, pen = swdist.count_nodes_allreduce(shard, k, w, tar, engine=swdist.HipEngine("host"))
        np.savez(out_path + f".r{rank}.npz", hash=h, n_tar=nt, n_neg=nn, penalty=pen)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_world1(tmp_path):
    """backend "nccl" (= RCCL): all_gather_object, three all_to_all_single with split sizes (int64 rows, int32
    ranks), the asynchronous all_gather -- issued for real at world size 1, result must equal the oracle."""
    import socket

    import torch.multiprocessing as mp
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    tar = [i % 2 == 0 for i in range(len(paths))]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "rccl.npz"
    mp.spawn(_rccl_worker, nprocs=1, args=(port, paths, 21, 200, tar, str(out)), join=True)
    got = np.load(out)
    ek, en, ee, eo, _ = oracle.build(paths, 21, 200)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(got["kmers"], ek) and np.array_equal(got["nodes"], en)
    assert np.array_equal(got["edges"], ee) and np.array_equal(got["record_offsets"], eo)
    assert np.array_equal(got["c_hash"], en["hash"]) and np.array_equal(got["c_tar"], en["n_tar"])       # the count-only path over RCCL
    assert np.array_equal(got["c_neg"], en["n_neg"]) and np.array_equal(got["c_pen"], en["penalty"])


def routed_tuple_exchange(paths, world, k, w, tar, packed=True, pairs=None, shards=None, sums_only=False, requests=False):
    """The tuple-exchange form with the all-to-all steps done by hand on ONE GPU (P shards, one after another).
    pairs (default: whenever the slices marked their ranks, as dist.py decides): adjacency as pair keys + candidate rows;
    else packed keys (when they fit) or {pair, assembly} rows.
    shards: ready-made Shard objects instead of FASTA paths; sums_only: return (sizes, checksums) of the concatenated
    result -- every slice's share at its offsets, added modulo 2^64 -- instead of the arrays (full-size runs).
    requests (pairs form): no job-wide rank -> hash table; the edge owners ask the node owners for the hashes of their edges'
    distinct endpoints (dist.hash_route "requests")."""
    eng = swdist.HipEngine()
    if shards is None:
        parts = swdist.partition_assemblies(len(paths), world)
        shards = [swdist.Shard(Batch.from_fasta(paths[a:b], n_cpu=2), a, len(paths)) for a, b in parts]
    assert len(shards) == world
    occs = [eng.sketch(s, k, w) for s in shards]
    offs = [s.batch.record_offsets() for s in shards]
    rec_base, glob, total = [], [np.zeros(1, np.uint32)], 0
    for o in offs:
        rec_base.append(total)
        glob.append(o[1:] + np.uint32(total))
        total += int(o[-1])
    record_offsets = np.concatenate(glob)
    nb, _ = swdist.hash_bounds(world)
    rows, perms, cnts = zip(*[eng.partition(occs[r], nb, rec_base[r]) for r in range(world)])
    cuts = [np.concatenate([[0], np.cumsum(c)]) for c in cnts]
    slices, ranks_back, kbase = [], [[None] * world for _ in range(world)], 0
    for owner in range(world):
        r_rows = torch.cat([rows[r][cuts[r][owner]:cuts[r][owner + 1]] for r in range(world)])
        ix, r_ranks = eng.slice_build(r_rows, kbase, record_offsets, tar)
        slices.append((ix, r_ranks))
        kbase += r_rows.shape[0]
    node_cnt = [s[0].sizes()[1] for s in slices]
    node_base = swdist.node_bases(node_cnt)
    total_nodes = node_base[-1]
    REP = 0x80000000
    marked = [eng.ranks_marked(s[0]) for s in slices]
    if pairs is None:
        pairs = all(marked)
    assert not pairs or all(marked)
    for owner in range(world):
        ix, r_ranks = slices[owner]
        if pairs:
            g = r_ranks                              # slice-local, repeat mark in bit 31: the sources re-base them in the library
        else:
            rr = r_ranks.to(torch.int64) & 0xFFFFFFFF
            g = (((rr & (REP - 1)) if marked[owner] else rr) + node_base[owner]).to(torch.int32)
        o = 0
        for r in range(world):
            c = int(cnts[r][owner])
            ranks_back[r][owner] = g[o:o + c]
            o += c
    pad = max(1, max(node_cnt))
    rb = swdist.rank_bounds(world, total_nodes)
    if pairs:
        table = torch.cat([eng.node_hash_part(s[0], pad) for s in slices])          # what all_gather_into_tensor leaves
        asm_bits = max(1, int(shards[0].n_assemblies_total).bit_length())
    else:
        rank_hash = torch.cat([eng.node_hashes(s[0]) for s in slices])
        n_bits = max(1, total_nodes.bit_length())
        asm_bits = swdist.adjacency_asm_bits(n_bits, shards[0].n_assemblies_total) if packed else 0
    adj = []
    for r in range(world):
        by_row = torch.cat(ranks_back[r]) if occs[r].n else torch.zeros(0, dtype=torch.int32, device=eng.gpu)
        if pairs:
            adj.append(eng.adjacency_pairs(occs[r], by_row, node_base, shards[r].first_assembly, rb))
        else:
            adj.append(eng.adjacency(occs[r], perms[r], by_row, n_bits, asm_bits, shards[r].first_assembly, rb))
    kmers, nodes, edges = [], [], []
    for owner in range(world):
        pieces, cpieces = [], []
        for r in range(world):
            a_rows, a_cnt = adj[r][0], adj[r][1]
            c = np.concatenate([[0], np.cumsum(a_cnt)])
            pieces.append(a_rows[c[owner]:c[owner + 1]])
            if pairs:
                cc = np.concatenate([[0], np.cumsum(adj[r][3])])
                cpieces.append(adj[r][2][cc[owner]:cc[owner + 1]])
        if pairs:
            assert all(a[4] == adj[0][4] for a in adj)        # every source derives the same key layout
            eng.slice_edges_pairs(slices[owner][0], torch.cat(pieces), torch.cat(cpieces), adj[0][4], rb[owner - 1] if owner else 0,
                                  asm_bits, None if requests else table, node_base, pad)
        else:
            eng.slice_edges(slices[owner][0], torch.cat(pieces), n_bits, asm_bits, rank_hash)
    if pairs and requests:
        # requests to node owner o: the pieces of all edge owners, in edge-owner order; the answers go back the same way
        asked = [eng.edge_hash_requests(slices[q][0], node_base) for q in range(world)]
        cuts_q = [np.concatenate([[0], np.cumsum(c)]) for _, c in asked]
        answers = [[None] * world for _ in range(world)]
        for o in range(world):
            got = torch.cat([asked[q][0][cuts_q[q][o]:cuts_q[q][o + 1]] for q in range(world)])
            ans = eng.node_hash_lookup(slices[o][0], got)
            at = 0
            for q in range(world):
                c = int(asked[q][1][o])
                answers[q][o] = ans[at:at + c]
                at += c
        for q in range(world):
            eng.edge_hash_attach(slices[q][0], torch.cat(answers[q]))
    for owner in range(world):
        if not sums_only:
            K, N, E = slices[owner][0].export()
            kmers.append(K); nodes.append(N); edges.append(E)
    if sums_only:
        sizes, sums = [0, 0, 0], [0, 0, 0]
        for ix, _ in slices:
            share = ix.checksums(*sizes)
            sums = [(a + b) % 2**64 for a, b in zip(sums, share)]
            sizes = [a + b for a, b in zip(sizes, ix.sizes())]
            ix.close()
        return tuple(sizes), tuple(sums)
    return np.concatenate(kmers), np.concatenate(nodes), np.concatenate(edges), record_offsets


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_routed_tuple_exchange_equals_single_batch(world):
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    tar = [i % 3 != 0 for i in range(len(paths))]
    for k, w in [(15, 20), (21, 200), (11, 5)]:
        ek, en, ee, eo, _ = oracle.build(paths, k, w)
        oracle.get_penalty(ek, en, eo, tar)
        # pair keys + candidate rows (what dist.py chooses); one packed 64-bit key per row; {key, assembly} rows
        for packed, pairs in ((True, None), (True, False), (False, False)):
            got = routed_tuple_exchange(paths, world, k, w, tar, packed, pairs)
            assert np.array_equal(got[0], ek) and np.array_equal(got[1], en), (world, k, w, packed, pairs)
            assert np.array_equal(got[2], ee) and np.array_equal(got[3], eo), (world, k, w, packed, pairs)
        got = routed_tuple_exchange(paths, world, k, w, tar, requests=True)       # hashes asked for, no job-wide table
        assert all(np.array_equal(a, b) for a, b in zip(got, (ek, en, ee, eo))), (world, k, w, "requests")


def test_routed_tuple_exchange_with_33_bit_ranks(monkeypatch):
    """BASELINE configs[4] at k >= 19 has ~5e9 distinct minimizers over 8 GPUs: global node ranks need 33 bits and an edge key
    holds rank_lo relative to its owner's range (31 bits at most) next to a 33-bit rank_hi.  SEQWIN_DIST_NODE_SPACING leaves
    625 M unused ranks behind every owner's nodes, so this small set walks through exactly those widths."""
    monkeypatch.setenv("SEQWIN_DIST_NODE_SPACING", str(625_000_000))
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    tar = [i % 3 != 0 for i in range(len(paths))]
    for k, w in [(15, 20), (21, 200)]:
        ek, en, ee, eo, _ = oracle.build(paths, k, w)
        oracle.get_penalty(ek, en, eo, tar)
        for requests in (False, True):
            got = routed_tuple_exchange(paths, 8, k, w, tar, requests=requests)
            assert np.array_equal(got[0], ek) and np.array_equal(got[1], en) and np.array_equal(got[2], ee) and np.array_equal(got[3], eo)
    # the key layout such a job gets: 33-bit rank_hi, rank_lo within 31 bits
    eng = swdist.HipEngine()
    shard = swdist.Shard(Batch.from_fasta(paths[:2], n_cpu=2), 0, len(paths))
    occ = eng.sketch(shard, 15, 20)
    rows, _, cnt = eng.partition(occ, swdist.hash_bounds(8)[0], 0)
    base = swdist.node_bases([1000] * 8)
    assert base[-1] > 2**32
    out = eng.adjacency_pairs(occ, torch.zeros(occ.n, dtype=torch.int32, device=eng.gpu), base, 0, swdist.rank_bounds(8, base[-1]))
    assert out[4] == (31, 33)
    # two owners cannot hold that many ranks in a 64-bit key: refused, not wrapped
    rows2, _, _ = eng.partition(occ, swdist.hash_bounds(2)[0], 0)
    with pytest.raises(RuntimeError, match="64-bit edge keys"):
        eng.adjacency_pairs(occ, torch.zeros(occ.n, dtype=torch.int32, device=eng.gpu), [0, 2**32, 2**33], 0,
                            swdist.rank_bounds(2, 2**33))


@pytest.mark.parametrize("case", ["few_tops_random_lows", "pairs_sharing_top", "ascending_lows", "random", "one_top_long_runs",
                                  "few_short_mixed_runs", "long_run_few_descents"])
def test_two_phase_hash_sort_repairs_shared_top_halves(case):
    """The node sort orders by the top 32 bits first and repairs runs whose low halves are out of order.  The slice
    builder takes arbitrary (hash, kmer) rows, so crafted hashes reach every branch of the repair: none needed,
    a sparse set of runs, and more descents than the list holds (whole-array fallback)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()))
    n = 300_000
    if case == "few_tops_random_lows":            # ~n/2 descents: more than the list capacity
        h = (rng.integers(0, 3, n, dtype=np.uint64) << np.uint64(32)) | rng.integers(0, 2**32, n, dtype=np.uint64)
    elif case == "pairs_sharing_top":             # 2000 tops, two full hashes each, heavily repeated
        tops = rng.integers(0, 2**32, 2000, dtype=np.uint64)
        lows = rng.integers(0, 2**32, (2000, 2), dtype=np.uint64)
        t = rng.integers(0, 2000, n)
        h = (tops[t] << np.uint64(32)) | lows[t, rng.integers(0, 2, n)]
    elif case == "ascending_lows":                # shared tops but already in order: nothing to repair
        h = (np.uint64(7) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    elif case == "one_top_long_runs":             # one shared top, three hashes, 100k occurrences each, interleaved
        h = (np.uint64(0xFFFFFFFF) << np.uint64(32)) | rng.permutation(np.repeat(np.array([5, 1, 3], np.uint64), n // 3))
        # plus a sprinkling of unrelated hashes so that most runs are clean
        h = np.concatenate([h, rng.integers(0, 2**63, n, dtype=np.uint64)])
        h = h[rng.permutation(len(h))]
    elif case == "few_short_mixed_runs":          # ~1 000 descents in runs of <= 2048: repaired in place, no host round trip
        tops = rng.integers(0, 2**32, 60, dtype=np.uint64)
        lows = rng.integers(0, 2**32, (60, 3), dtype=np.uint64)
        t = rng.integers(0, 60, 4000)
        mixed = (tops[t] << np.uint64(32)) | lows[t, rng.integers(0, 3, 4000)]
        h = np.concatenate([mixed, rng.integers(0, 2**63, n, dtype=np.uint64)])
        h = h[rng.permutation(len(h))]
    elif case == "long_run_few_descents":         # one run longer than the in-place limit with three descents
        run = (np.uint64(9) << np.uint64(32)) | np.sort(rng.integers(0, 2**32, 6000, dtype=np.uint64))
        run[[100, 3000, 5999]] = (np.uint64(9) << np.uint64(32)) | np.uint64(1)
        h = np.concatenate([run, rng.integers(0, 2**63, n, dtype=np.uint64)])    # arrival order = run order
    else:
        h = rng.integers(0, 2**64, n, dtype=np.uint64)
    n = len(h)
    kmer = np.arange(n, dtype=np.uint64) | (np.uint64(0) << np.uint64(32))     # record 0, pos = arrival index
    rows = torch.from_numpy(np.stack([h, kmer], axis=1).view(np.int64))
    eng = swdist.HipEngine()
    ix, ranks = eng.slice_build(rows.to(eng.gpu), 0, np.array([0, 1], np.uint32), None)
    K, N, E = ix.export()
    order = np.argsort(h, kind="stable")
    assert np.array_equal(K["pos"], order.astype(np.uint32))
    uh, start, cnt = np.unique(h[order], return_index=True, return_counts=True)
    assert np.array_equal(N["hash"], uh) and np.array_equal(N["start"], start.astype(np.uint64))
    assert np.array_equal(N["stop"], (start + cnt).astype(np.uint64))
    got = ranks.cpu().numpy().view(np.uint32)
    node_of = np.searchsorted(uh, h)
    assert np.array_equal(got & np.uint32(0x7FFFFFFF), node_of.astype(np.uint32))
    # bit 31 (sw_index_ranks_marked): the row's node occurs more than once in the row's assembly (here: one assembly)
    assert eng.ranks_marked(ix)
    assert np.array_equal(got >> np.uint32(31), (cnt[node_of] > 1).astype(np.uint32))


@pytest.mark.parametrize("workload,extra", [("tiny", []), ("salmonella500", ["--scaling", "strong", "--genomes", "48"])])
def test_bench_strong_scaling_reproduces_n1_checksums(workload, extra):
    """bench.py --gpus 2 (strong scaling: the SAME genomes sharded by build.cpp:350-356's partition, every collective of
    seqwin_amd/dist.py issued for real -- over gloo with host staging, two ranks sharing this box's one GPU) must print
    the checksums and counts of the single-GPU run: shard-count invariance as the driver's SCALE runs see it."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    base = [str(root / "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra

    def run(cmd, env=None):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    one = run([sys.executable] + base)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import os
    env = dict(os.environ, SEQWIN_BENCH_BACKEND="gloo", SEQWIN_BENCH_ALLOW_SHARED_GPU="1")   # (two ranks on the one card: a rehearsal)
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base + ["--gpus", "2"], env)
    assert two["dist"]["distinct_gpus"] == 1 and two["dist"]["collectives_checked"]
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["genomes"] == one["config"]["genomes"] and two["config"]["genomes_per_gpu"] * 2 == one["config"]["genomes"]
    assert two["counts"] == one["counts"] and two["checksums"] == one["checksums"]


@pytest.mark.parametrize("force,allow,needle", [("", "", "distinct GPU"), ("peer", "SEQWIN_BENCH_ALLOW_SHARED_GPU", "no peer access"),
                                                 ("collectives", "SEQWIN_BENCH_ALLOW_SHARED_GPU", "collective self-check failed")])
def test_bench_preflight_refuses_to_time_a_run_that_is_not_a_scaling_point(force, allow, needle):
    """bench.py --gpus N times nothing unless the N ranks drive N distinct GPUs, every ordered pair has peer access and the
    collectives deliver messages of the real round size (the day a multi-GPU node appears, SCALE is one shot).
    Each refusal is forced here on the one card: two ranks share it (refused unless allowed), then the peer probe and the
    collective check are made to fail.  A refused run prints ONE JSON line with value null and the reason, and exits with 3."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SEQWIN_BENCH_BACKEND="gloo")
    env.pop("SEQWIN_BENCH_ALLOW_SHARED_GPU", None)
    if force:
        env["SEQWIN_BENCH_PREFLIGHT_FORCE"] = force
    if allow:
        env[allow] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(root / "bench.py"), "--workload", "tiny", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] is None and needle in d["refused"] and d["n_gpus"] == 2 and len(d["preflight"]["ranks"]) == 2
    assert "NOT TIMED" in out.stderr
