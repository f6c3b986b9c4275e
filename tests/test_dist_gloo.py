"""CPU, world_size 2 over gloo: the multi-GPU choreography of seqwin_amd/dist.py (assembly partition,
record re-basing, hash-range all-to-all of occurrence and edge rows, slice merge, gather).

The per-rank compute is supplied by a numpy engine built on the oracle (test infrastructure), so what is
under test here is exactly the code that runs between the kernels on a multi-GPU node; the HIP engine
plugs into the same functions.  The merged result must equal the single-process result bit for bit
(shard-count invariance, reference tests/smoke/test_graph.py:67-127)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import GOLDEN
from seqwin_amd import dist as swdist

U64 = np.uint64


class NumpyEngine:
    device = torch.device("cpu")

    def local_index(self, shard, k, w):
        kmers, nodes, edges, offs, _ = oracle.build(shard.batch, k, w)
        return dict(kmers=kmers, nodes=nodes, edges=edges, offs=offs)

    def record_offsets(self, shard):
        return oracle.build(shard.batch, 3, 1)[3] if False else self._offs

    def sizes(self, ix):
        return len(ix["kmers"]), len(ix["nodes"]), len(ix["edges"])

    def timings(self, ix):
        return {}

    def splits(self, ix, node_bounds, edge_bounds):
        nodes, edges = ix["nodes"], ix["edges"]
        osp, esp = [], []
        for b in node_bounds:
            i = int(np.searchsorted(nodes["hash"], U64(b), side="left"))
            osp.append(int(nodes["start"][i]) if i < len(nodes) else len(ix["kmers"]))
        for b in edge_bounds:
            esp.append(int(np.searchsorted(edges["first"], U64(b), side="left")))
        return osp, esp

    def occ_rows(self, ix, rec_offset):
        nodes, kmers = ix["nodes"], ix["kmers"]
        h = np.repeat(nodes["hash"], (nodes["stop"] - nodes["start"]).astype(np.int64))
        km = kmers["pos"].astype(U64) | ((kmers["record_idx"].astype(U64) + U64(rec_offset)) << U64(32))
        return torch.from_numpy(np.stack([h, km], axis=1).view(np.int64).copy())

    def edge_rows(self, ix):
        return torch.from_numpy(ix["edges"].view(U64).reshape(-1, 3).view(np.int64).copy())

    def merge(self, occ_rows, edge_rows, kmer_base, record_offsets, is_targets):
        occ = occ_rows.numpy().view(U64)
        order = np.argsort(occ[:, 0], kind="stable")
        h, km = occ[order, 0], occ[order, 1]
        kmers = np.empty(len(km), oracle.KMER_DTYPE)
        kmers["pos"] = (km & U64(0xFFFFFFFF)).astype(np.uint32)
        kmers["record_idx"] = (km >> U64(32)).astype(np.uint32)
        uh, start = np.unique(h, return_index=True)
        nodes = np.zeros(len(uh), oracle.NODE_DTYPE)
        nodes["hash"] = uh
        nodes["start"] = start
        nodes["stop"] = np.append(start[1:], len(h))
        if is_targets is not None and len(nodes):
            oracle.get_penalty(kmers, nodes, record_offsets, is_targets)
        nodes["start"] += U64(kmer_base)
        nodes["stop"] += U64(kmer_base)
        e = edge_rows.numpy().view(U64)
        o = np.lexsort((e[:, 1], e[:, 0]))
        e = e[o]
        edges = np.zeros(0, oracle.EDGE_DTYPE)
        if len(e):
            head = np.ones(len(e), bool)
            head[1:] = (e[1:, 0] != e[:-1, 0]) | (e[1:, 1] != e[:-1, 1])
            idx = np.nonzero(head)[0]
            edges = np.zeros(len(idx), oracle.EDGE_DTYPE)
            edges["first"], edges["second"] = e[idx, 0], e[idx, 1]
            edges["weight"] = np.add.reduceat(e[:, 2], idx)
        return dict(kmers=kmers, nodes=nodes, edges=edges)

    def export(self, ix):
        return ix["kmers"], ix["nodes"], ix["edges"]

    def checksums(self, ix, kmer_base=0, node_base=0, edge_base=0):
        from seqwin_amd.device import host_checksums
        return host_checksums(ix["kmers"], ix["nodes"], ix["edges"], kmer_base, node_base, edge_base)

    # ---- tuple-exchange form (same interface as seqwin_amd.dist.HipEngine) -------------------------------
    def sketch(self, shard, k, w):
        from types import SimpleNamespace
        hs, km, rec_asm, rec = [np.zeros(0, U64)], [np.zeros(0, U64)], [], 0
        for a, path in enumerate(shard.batch):
            for _id, seq in oracle.read_fasta(path):
                _, oh, pos = oracle.minimize(seq, k, w)
                hs.append(oh)
                km.append(pos.astype(U64) | (U64(rec) << U64(32)))
                rec_asm.append(a)
                rec += 1
        h = np.concatenate(hs)
        return SimpleNamespace(n=len(h), hash=h, kmer=np.concatenate(km), rec_asm=np.array(rec_asm, np.int64), sketch_ms=0.0)

    def partition(self, occ, bounds, rec_offset):
        b = np.array(bounds, dtype=U64)
        owner = np.searchsorted(b, occ.hash, side="right")
        perm = np.argsort(owner, kind="stable")
        rows = np.stack([occ.hash[perm], occ.kmer[perm] + (U64(rec_offset) << U64(32))], axis=1)
        counts = np.bincount(owner, minlength=len(bounds) + 1).tolist()
        occ.perm, occ.counts = perm, counts   # (the HIP engine's handle remembers its partition the same way)
        return torch.from_numpy(rows.view(np.int64).copy()), torch.from_numpy(perm.astype(np.int32)), counts

    def slice_build(self, rows, kmer_base, record_offsets, is_targets):
        ix = self.merge(rows, torch.zeros((0, 3), dtype=torch.int64), kmer_base, record_offsets, is_targets)
        r = rows.numpy().view(U64)
        ranks = np.searchsorted(ix["nodes"]["hash"], r[:, 0]).astype(np.uint32)
        if self.mark_repeats and len(r):
            # bit 31: the row's node occurs more than once in the row's assembly
            asm = np.searchsorted(np.asarray(record_offsets, np.uint64), r[:, 1] >> U64(32), side="right") - 1
            both = np.stack([ranks.astype(np.int64), asm.astype(np.int64)], axis=1)
            _, inv, cnt = np.unique(both, axis=0, return_inverse=True, return_counts=True)
            ranks = ranks | np.where(cnt[inv.ravel()] > 1, np.uint32(0x80000000), np.uint32(0))
        ix["marked"] = bool(self.mark_repeats)
        return ix, torch.from_numpy(ranks.view(np.int32).copy())

    mark_repeats = True

    def ranks_marked(self, ix):
        return ix.get("marked", False)

    def adjacency_pairs(self, occ, ranks_by_row, node_base, asm_base, rank_bounds):
        # slice-local ranks (+ repeat mark) come back in partitioned-row order; the owner of row j follows from the partition
        w = np.zeros(occ.n, U64)
        w[occ.perm] = ranks_by_row.numpy().view(np.uint32).astype(U64)
        own = np.zeros(occ.n, np.int64)
        own[occ.perm] = np.repeat(np.arange(len(occ.counts)), occ.counts)
        rank = (w & U64(0x7FFFFFFF)) + np.asarray(node_base, U64)[own]
        rep = (w >> U64(31)).astype(bool)
        total = int(node_base[-1])
        lo_base = np.array([0] + [int(b) for b in rank_bounds], dtype=U64)
        ends = [int(b) for b in rank_bounds] + [total]
        widest = max(e - int(b) for e, b in zip(ends, lo_base))
        hi_bits, lo_bits = max(1, total.bit_length()), max(1, int(widest).bit_length())
        assert lo_bits + hi_bits <= 64
        rec = (occ.kmer >> U64(32)).astype(np.int64)
        ok = rec[1:] == rec[:-1] if occ.n > 1 else np.zeros(0, bool)
        u, v = np.minimum(rank[:-1], rank[1:])[ok], np.maximum(rank[:-1], rank[1:])[ok]
        owner = np.searchsorted(np.array(rank_bounds, dtype=U64), u, side="right")
        key = ((u - lo_base[owner]) << U64(hi_bits)) | v
        asm = (occ.rec_asm[rec[:-1][ok]] + asm_base).astype(U64)
        cand = (rep[:-1] | rep[1:])[ok] if occ.n > 1 else np.zeros(0, bool)
        p = np.argsort(owner, kind="stable")
        co = owner[cand]
        cp = np.argsort(co, kind="stable")
        crows = np.stack([key[cand][cp], asm[cand][cp]], axis=1) if cand.any() else np.zeros((0, 2), U64)
        return (torch.from_numpy(key[p].view(np.int64).copy()), np.bincount(owner, minlength=len(rank_bounds) + 1).tolist(),
                torch.from_numpy(crows.view(np.int64).copy()), np.bincount(co, minlength=len(rank_bounds) + 1).tolist(),
                (lo_bits, hi_bits))

    def slice_edges_pairs(self, ix, keys, cand, key_bits, lo_base, asm_bits, rank_hash, node_base, pad):
        k = keys.numpy().view(U64)
        base = np.asarray(node_base, U64)
        ix["edges_hold_ranks"] = rank_hash is None
        if rank_hash is None:
            def hash_of(rank):   # no table: the edges keep global ranks until edge_hash_attach
                return rank
        else:
            table = rank_hash.numpy().view(U64)

            def hash_of(rank):   # owner o's hashes sit at table[o * pad ...]
                o = np.searchsorted(base[1:-1], rank, side="right")
                return table[(o.astype(U64) * U64(pad) + (rank - base[o])).astype(np.int64)]

        edges = np.zeros(0, oracle.EDGE_DTYPE)
        if len(k):
            uk, cnt = np.unique(k, return_counts=True)
            w = cnt.astype(np.int64)
            c = cand.numpy().view(U64)
            if len(c):   # records that repeat their (pair, assembly) do not count
                rows, rc = np.unique(c, axis=0, return_counts=True)
                np.subtract.at(w, np.searchsorted(uk, rows[:, 0]), rc - 1)
            hi_bits = key_bits[1]
            edges = np.zeros(len(uk), oracle.EDGE_DTYPE)
            edges["first"] = hash_of((uk >> U64(hi_bits)) + U64(lo_base))
            edges["second"] = hash_of(uk & U64((1 << hi_bits) - 1))
            edges["weight"] = w.astype(np.uint64)
        ix["edges"] = edges

    def edge_hash_requests(self, ix, node_base):
        assert ix["edges_hold_ranks"]
        base = np.asarray(node_base, U64)
        e = ix["edges"]
        ranks = np.stack([e["first"], e["second"]], axis=1).ravel()          # slot = edge * 2 + side
        uniq, inv = np.unique(ranks, return_inverse=True)
        owner = np.searchsorted(base[1:-1], uniq, side="right")
        ix["hash_job"] = inv.ravel()
        local = (uniq - base[owner]).astype(np.uint32)
        return torch.from_numpy(local.view(np.int32).copy()), np.bincount(owner, minlength=len(node_base) - 1).tolist()

    def node_hash_lookup(self, ix, local_ranks):
        r = local_ranks.numpy().view(np.uint32).astype(np.int64)
        return torch.from_numpy(ix["nodes"]["hash"][r].view(np.int64).copy())

    def edge_hash_attach(self, ix, replies):
        h = replies.numpy().view(U64)[ix.pop("hash_job")].reshape(-1, 2)
        ix["edges"]["first"], ix["edges"]["second"] = h[:, 0], h[:, 1]
        ix["edges_hold_ranks"] = False

    def node_hash_part(self, ix, pad):
        out = np.zeros(pad, U64)
        out[:len(ix["nodes"])] = ix["nodes"]["hash"]
        return torch.from_numpy(out.view(np.int64).copy())

    def node_hashes(self, ix):
        return torch.from_numpy(ix["nodes"]["hash"].view(np.int64).copy())

    def adjacency(self, occ, perm, ranks_by_row, n_bits, asm_bits, asm_base, rank_bounds):
        rank = np.zeros(occ.n, U64)
        rank[perm.numpy()] = ranks_by_row.numpy().view(np.uint32).astype(U64)
        rec = (occ.kmer >> U64(32)).astype(np.int64)
        ok = rec[1:] == rec[:-1] if occ.n > 1 else np.zeros(0, bool)
        u, v = rank[:-1][ok], rank[1:][ok]
        key = (np.minimum(u, v) << U64(n_bits)) | np.maximum(u, v)
        asm = (occ.rec_asm[rec[:-1][ok]] + asm_base).astype(U64)
        kb = np.array([b << n_bits for b in rank_bounds], dtype=U64)
        owner = np.searchsorted(kb, key, side="right")
        p = np.argsort(owner, kind="stable")
        if asm_bits:
            rows = (key[p] << U64(asm_bits)) | asm[p]
        else:
            rows = np.stack([key[p], asm[p]], axis=1) if len(key) else np.zeros((0, 2), U64)
        return torch.from_numpy(rows.view(np.int64).copy()), np.bincount(owner, minlength=len(rank_bounds) + 1).tolist()

    def slice_edges(self, ix, adj_rows, n_bits, asm_bits, rank_hash):
        r = adj_rows.numpy().view(U64)
        if asm_bits:
            r = np.stack([r >> U64(asm_bits), r & U64((1 << asm_bits) - 1)], axis=1)
        table = rank_hash.numpy().view(U64)
        edges = np.zeros(0, oracle.EDGE_DTYPE)
        if len(r):
            o = np.argsort(r[:, 0], kind="stable")
            key, asm = r[o, 0], r[o, 1]
            head = np.ones(len(key), bool)
            head[1:] = key[1:] != key[:-1]
            change = head.copy()
            change[1:] |= asm[1:] != asm[:-1]
            idx = np.nonzero(head)[0]
            edges = np.zeros(len(idx), oracle.EDGE_DTYPE)
            edges["first"] = table[(key[idx] >> U64(n_bits)).astype(np.int64)]
            edges["second"] = table[(key[idx] & U64((1 << n_bits) - 1)).astype(np.int64)]
            edges["weight"] = np.add.reduceat(change.astype(np.uint64), idx)
        ix["edges"] = edges

    def free_occ(self, occ):
        pass


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, paths, k, w, tar, out_path, mode):
    os.environ["SEQWIN_DIST_SELFCHECK_MB"] = "1"   # (the start-up self-check of the collectives, at a CPU-sized message)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, end = swdist.partition_assemblies(len(paths), world)[rank]
        eng = NumpyEngine()
        if mode.endswith("+rounds"):               # every exchange in rounds of at most 48 bytes per peer (three 16-byte rows)
            swdist._MSG_LIMIT, mode = 48, mode[:-len("+rounds")]
        eng.mark_repeats = mode != "tuples_rows"   # "tuples": adjacency in its pairs form; "tuples_rows": {pair, assembly} rows
        if mode == "tuples_requests":              # ... and the edge owners ask the node owners for hashes (no job-wide table)
            os.environ["SEQWIN_DIST_HASH_ROUTE"] = "requests"
        mine = paths[start:end]
        eng._offs = oracle.build(mine, k, w)[3]
        build = swdist.build_sharded_index if mode.startswith("tuples") else swdist.build_sharded_index_merge
        sharded = build(swdist.Shard(mine, start, len(paths)), k, w, tar, engine=eng)
        sums = sharded.global_checksums()   # every rank: shares of the slices add up to the checksums of the whole
        full = sharded.gather(0)
        assert (full is None) == (rank != 0)
        if rank == 0:
            np.savez(out_path, kmers=full[0], nodes=full[1], edges=full[2], record_offsets=full[3],
                     sums=np.array(sums, np.uint64))
        else:
            np.save(out_path + f".sums{rank}.npy", np.array(sums, np.uint64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tuples", "tuples_requests", "tuples_rows", "merge"])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", ["smoke", "pan", "edge"])
def test_sharded_build_equals_single(tmp_path, world, case, mode):
    if case == "smoke":
        s = GOLDEN / "smoke"
        paths = [s / "targets/target-1.fasta", s / "targets/target-2.fasta",
                 s / "non-targets/non-target-1.fasta", s / "non-targets/non-target-2.fasta"]
        k, w = 7, 10
    elif case == "pan":
        paths = sorted((GOLDEN / "synth").glob("pan_*.fa"))
        k, w = 15, 20
    else:
        paths = sorted((GOLDEN / "synth").glob("edge_*"))
        k, w = 11, 5
    paths = [str(p) for p in paths]
    tar = [i % 2 == 0 for i in range(len(paths))]
    out = tmp_path / "merged.npz"
    mp.spawn(_worker, nprocs=world, args=(world, _free_port(), paths, k, w, tar, str(out), mode), join=True)
    got = np.load(out)
    ek, en, ee, eo, _ = oracle.build(paths, k, w)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(got["kmers"], ek) and np.array_equal(got["nodes"], en)
    assert np.array_equal(got["edges"], ee) and np.array_equal(got["record_offsets"], eo)
    # shard-count invariance without gathering: the all-reduced checksums are those of the single-process result
    from seqwin_amd.device import host_checksums
    exp_sums = np.array(host_checksums(ek, en, ee), np.uint64)
    assert np.array_equal(got["sums"], exp_sums)
    for r in range(1, world):
        assert np.array_equal(np.load(str(out) + f".sums{r}.npy"), exp_sums)


def _counts_worker(rank, world, port, paths, k, w, tar, out_path):
    os.environ["SEQWIN_DIST_SELFCHECK_MB"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, end = swdist.partition_assemblies(len(paths), world)[rank]
        eng = NumpyEngine()
        mine = paths[start:end]
        eng._offs = oracle.build(mine, k, w)[3]
        h, nt, nn, pen = swdist.count_nodes_allreduce(swdist.Shard(mine, start, len(paths)), k, w, tar, engine=eng)
        np.savez(out_path + f".r{rank}.npz", hash=h, n_tar=nt, n_neg=nn, penalty=pen)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_count_only_allreduce_equals_the_merged_graph(tmp_path, world):
    """SURVEY 8e C2 / the north_star's "single RCCL reduce": every rank counts its own shard, one dictionary, one all_reduce of the
    dense count vectors -- hash, n_tar, n_neg and the f64 penalty of every node of the whole job on every rank, bit for bit what
    get_penalty leaves in the merged graph (shards with only targets, only non-targets or no assembly at all included)."""
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa")) + sorted((GOLDEN / "synth").glob("edge_*"))]
    k, w = 15, 20
    tar = [i < 3 for i in range(len(paths))]                       # the first shard(s) hold targets only
    out = str(tmp_path / "counts")
    mp.spawn(_counts_worker, nprocs=world, args=(world, _free_port(), paths, k, w, tar, out), join=True)
    ek, en, ee, eo, _ = oracle.build(paths, k, w)
    oracle.get_penalty(ek, en, eo, tar)
    for r in range(world):
        got = np.load(out + f".r{r}.npz")
        assert np.array_equal(got["hash"], en["hash"]) and np.array_equal(got["n_tar"], en["n_tar"]), r
        assert np.array_equal(got["n_neg"], en["n_neg"]) and np.array_equal(got["penalty"], en["penalty"]), r   # f64, tolerance 0


def test_count_only_path_single_process_and_validation():
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa"))]
    k, w = 11, 5
    tar = [i % 2 == 0 for i in range(len(paths))]
    eng = NumpyEngine()
    eng._offs = oracle.build(paths, k, w)[3]
    h, nt, nn, pen = swdist.count_nodes_allreduce(swdist.Shard(paths, 0, len(paths)), k, w, tar, engine=eng)
    ek, en, ee, eo, _ = oracle.build(paths, k, w)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(h, en["hash"]) and np.array_equal(nt, en["n_tar"]) and np.array_equal(nn, en["n_neg"])
    assert np.array_equal(pen, en["penalty"])
    for bad in ([True] * len(paths), [False] * len(paths), tar[:-1]):
        with pytest.raises(ValueError):
            swdist.count_nodes_allreduce(swdist.Shard(paths, 0, len(paths)), k, w, bad, engine=eng)


def test_collective_self_check_catches_a_truncating_transport(monkeypatch):
    """What RCCL 2.26 did at world size 1 in round 3 -- the first half of a large all_to_all_single message delivered, the
    rest of the output left untouched, no error -- must fail the start-up check instead of the graph."""
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(_free_port()))
    monkeypatch.setenv("SEQWIN_DIST_SELFCHECK_MB", "1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        swdist._checked_groups.clear()
        swdist.check_collectives(torch.device("cpu"))                # a sound transport passes (and is remembered)
        assert swdist._checked_groups
        swdist._checked_groups.clear()
        real = dist.all_to_all_single

        def half(out, inp, *a, **kw):
            tmp = torch.empty_like(out)
            real(tmp, inp, *a, **kw)
            out[:out.numel() // 2] = tmp[:out.numel() // 2]
        monkeypatch.setattr(dist, "all_to_all_single", half)
        with pytest.raises(RuntimeError, match="self-check failed"):
            swdist.check_collectives(torch.device("cpu"))
        assert not swdist._checked_groups
    finally:
        dist.destroy_process_group()


def _skewed_exchange_worker(rank, world, port, matrix, cap_rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        swdist._MSG_LIMIT = cap_rows * 16
        counts = matrix[rank]
        # row (src, dst, i) -> [src << 40 | dst << 20 | i, its negation]; an empty send keeps its (0, 2) shape
        rows = torch.tensor([[(rank << 40) | (d << 20) | i, -((rank << 40) | (d << 20) | i)] for d in range(world)
                             for i in range(counts[d])], dtype=torch.int64).reshape(-1, 2)
        got, recv, m = swdist._exchange_rows(rows, counts, torch.device("cpu"), None)
        assert m == matrix and recv == [matrix[s][rank] for s in range(world)]
        exp = [[(s << 40) | (rank << 20) | i, -((s << 40) | (rank << 20) | i)] for s in range(world) for i in range(matrix[s][rank])]
        assert got.tolist() == exp, (rank, got.shape)
        # and back along the transposed matrix (the way the node ranks and the hash replies travel)
        back = torch.empty((sum(counts), 2), dtype=torch.int64)
        swdist._all_to_all_rows(back, got, counts, recv, None, max(max(r) for r in m))
        assert torch.equal(back, rows)
        open(out_path + f".ok{rank}", "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("matrix", [[[5, 5, 5], [5, 5, 250], [5, 5, 5]], [[0, 0, 0], [0, 0, 301], [7, 0, 0]],
                                    [[100, 100, 100], [100, 100, 100], [100, 100, 101]]])
def test_rounds_follow_the_job_wide_maximum(tmp_path, matrix):
    """Skewed counts with a cap of 100 rows per message: one pair of ranks needs three rounds, the others one (or none -- a
    rank that sends nothing at all).  Every rank must issue the same number of collectives (decided from local
    counts, rank 0 did one call and ranks 1 and 2 three, and gloo timed out)."""
    out = str(tmp_path / "x")
    mp.spawn(_skewed_exchange_worker, nprocs=3, args=(3, _free_port(), matrix, 100, out), join=True)
    assert all(os.path.exists(out + f".ok{r}") for r in range(3))


@pytest.mark.parametrize("mode", ["tuples+rounds", "tuples_requests+rounds", "tuples_rows+rounds"])
def test_sharded_build_in_message_rounds(tmp_path, mode):
    """The whole choreography with every per-peer message cut into rounds of three rows (what runs above 256 MiB per peer:
    two GPUs at 15 000 genomes, configs[4])."""
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa"))]
    k, w = 15, 20
    tar = [i % 2 == 0 for i in range(len(paths))]
    out = tmp_path / "merged.npz"
    mp.spawn(_worker, nprocs=3, args=(3, _free_port(), paths, k, w, tar, str(out), mode), join=True)
    got = np.load(out)
    ek, en, ee, eo, _ = oracle.build(paths, k, w)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(got["kmers"], ek) and np.array_equal(got["nodes"], en)
    assert np.array_equal(got["edges"], ee) and np.array_equal(got["record_offsets"], eo)


@pytest.mark.parametrize("mode", ["tuples", "tuples_requests"])
def test_eight_ranks_with_empty_shards(tmp_path, mode):
    """World size 8 -- what `bench.py --gpus 8` runs -- on six assemblies: two ranks hold no assembly at all, some hash ranges
    and some edge ranges own nothing; every collective still has to line up (split sizes of zero, empty requests)."""
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa"))]
    assert len(paths) == 6
    k, w = 15, 20
    tar = [i % 2 == 0 for i in range(len(paths))]
    out = tmp_path / "merged.npz"
    mp.spawn(_worker, nprocs=8, args=(8, _free_port(), paths, k, w, tar, str(out), mode), join=True)
    got = np.load(out)
    ek, en, ee, eo, _ = oracle.build(paths, k, w)
    oracle.get_penalty(ek, en, eo, tar)
    assert np.array_equal(got["kmers"], ek) and np.array_equal(got["nodes"], en)
    assert np.array_equal(got["edges"], ee) and np.array_equal(got["record_offsets"], eo)
    from seqwin_amd.device import host_checksums
    assert np.array_equal(got["sums"], np.array(host_checksums(ek, en, ee), np.uint64))


def test_partition_formula():
    # cpp/src/seqwin/build.cpp:350-356
    assert swdist.partition_assemblies(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert swdist.partition_assemblies(4, 8)[:5] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 4)]
    assert swdist.partition_assemblies(0, 2) == [(0, 0), (0, 0)]
    for n in range(0, 40):
        for p in range(1, 9):
            parts = swdist.partition_assemblies(n, p)
            assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_single_process_tuple_form_equals_oracle():
    paths = [str(p) for p in sorted((GOLDEN / "synth").glob("pan_*.fa"))]
    eng = NumpyEngine()
    eng._offs = oracle.build(paths, 15, 20)[3]
    tar = [True, False, True, False, True, False]
    sharded = swdist.build_sharded_index(swdist.Shard(paths, 0, len(paths)), 15, 20, tar, engine=eng)
    ek, en, ee, eo, _ = oracle.build(paths, 15, 20)
    oracle.get_penalty(ek, en, eo, tar)
    k, n, e = sharded.export()
    assert np.array_equal(k, ek) and np.array_equal(n, en) and np.array_equal(e, ee)
    os.environ["SEQWIN_DIST_HASH_ROUTE"] = "requests"      # hashes asked for instead of looked up in the table
    try:
        sharded = swdist.build_sharded_index(swdist.Shard(paths, 0, len(paths)), 15, 20, tar, engine=eng)
    finally:
        del os.environ["SEQWIN_DIST_HASH_ROUTE"]
    k, n, e = sharded.export()
    assert np.array_equal(k, ek) and np.array_equal(n, en) and np.array_equal(e, ee)
    assert swdist.hash_route(10**6) == "table" and swdist.hash_route(5 * 10**9) == "requests"     # 8 MB / 40 GB of table
    assert swdist.hash_route(8 * 68 * 10**6) == "requests"    # 8 owners padded to the largest slice: 4.35 GB though the nodes are fewer


def test_rank_bounds():
    for p in (1, 2, 4, 8):
        for total in (0, 1, 7, 1000, 3_831_468, 2**32 - 2):
            rb = swdist.rank_bounds(p, total)
            assert len(rb) == p - 1 and rb == sorted(rb) and all(0 <= b <= total for b in rb)


def test_hash_bounds_are_monotone_and_balanced():
    for p in (1, 2, 3, 4, 8):
        nb, eb = swdist.hash_bounds(p)
        assert len(nb) == len(eb) == p - 1
        assert nb == sorted(nb) and eb == sorted(eb) and all(0 < b < 2**64 for b in nb + eb)
    rng = np.random.default_rng(0)
    u, v = rng.random(200000), rng.random(200000)
    first = np.minimum(u, v)
    nb, eb = swdist.hash_bounds(8)
    cnt = np.histogram(first, bins=[0] + [b / 2**64 for b in eb] + [1])[0]
    assert cnt.max() / cnt.min() < 1.15     # quantile splitters balance min(u, v)


def test_adjacency_key_width_and_bounds():
    """Packed adjacency rows (rank_lo, rank_hi, assembly) are used exactly when they fit 64 bits; owner boundaries are
    monotone and cover the whole range."""
    assert swdist.adjacency_asm_bits(22, 512) == 10            # 2*22 + 10 <= 64 (bit_length(512) = 10)
    assert swdist.adjacency_asm_bits(27, 15000) == 0           # 54 + 14 > 64: {pair, assembly} rows
    assert swdist.adjacency_asm_bits(1, 1) == 1
    assert swdist.adjacency_asm_bits(32, 1) == 0
    for world in (1, 2, 3, 8, 16):
        hb = swdist.hash_bounds(world)
        for bounds in hb if isinstance(hb, tuple) else (hb,):
            assert len(bounds) == world - 1 and list(bounds) == sorted(bounds)
        rb = swdist.rank_bounds(world, 1_000_003)
        assert len(rb) == world - 1 and list(rb) == sorted(rb) and all(0 <= x <= 1_000_003 for x in rb)
    parts = swdist.partition_assemblies(10, 4)                  # build.cpp:350-356: the remainder goes to the first workers
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_bench_preflight_counts_distinct_gpus():
    """bench.py's pre-flight (refuses to time N ranks that are not on N GPUs): the count of distinct GPUs from the ranks' records.  Two
    ranks on one card are one GPU however they see it; cards whose runtime reports equal or no UUIDs still count apart by bus id or
    device index -- a good 8-GPU node must not be refused."""
    import sys
    sys.path.insert(0, str(GOLDEN.parent.parent))
    from bench import distinct_gpus
    node = [{"host": "h", "uuid": f"u{i}", "pci_bus_id": 0x10 + i, "device": i, "visible_devices": 8} for i in range(8)]
    assert distinct_gpus(node) == 8
    assert distinct_gpus([dict(r, uuid="") for r in node]) == 8                       # no UUIDs: bus ids / indices
    assert distinct_gpus([dict(r, uuid="same", pci_bus_id=None) for r in node]) == 8  # equal UUIDs, no bus id: indices
    masked = [dict(r, device=0, visible_devices=1) for r in node]                    # one visible device per rank (HIP_VISIBLE_DEVICES)
    assert distinct_gpus(masked) == 8
    assert distinct_gpus([node[0], dict(node[0], rank=1)]) == 1                       # two ranks on one card
    assert distinct_gpus([masked[3], dict(masked[3], pid=2)]) == 1
    assert distinct_gpus(node + [dict(node[0], host="other")]) == 9                   # the same numbering on another host
    assert distinct_gpus([{"rank": 0, "host": "h", "error": "x"}, {"rank": 1, "host": "h", "error": "x"}]) == 1   # nothing known: refuse
