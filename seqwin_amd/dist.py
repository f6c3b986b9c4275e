"""One process per GPU: shard the assemblies, build partial graphs, merge them over RCCL.

This is the multi-GPU form of the reference's thread partition + ``merge_thread_graphs``
(cpp/src/seqwin/build.cpp:350-367, cpp/src/seqwin/build_internals.cpp:295-392):

* assemblies are split into contiguous ranges with the reference's formula (``partition_assemblies``);
  GPU g owns the range of "thread g", so global record_idx = local + prefix of the record counts;
* ``build_sharded_index`` (tuple exchange, what bench.py times): every GPU sketches its shard; the
  (out_hash, pos|record) tuples are exchanged by hash range with ``all_to_all_single`` (backend "nccl" is
  RCCL over xGMI; a direct all-to-all uses all seven links of a GPU at once, which is why it is preferred
  over a ring reduce here); the owner of a hash range sorts its tuples ONCE (stable, so ties keep
  source-rank = record order), run-lengths them into nodes / kmers, counts target / non-target
  assemblies, and returns the node rank of every tuple to its source; sources turn consecutive
  minimizers into (rank_lo, rank_hi, assembly) rows, which are exchanged by rank range (quantile-shaped
  splitters, because min(u, v) is skewed) and reduced to weighted edges by their owner;
* ``build_sharded_index_merge`` (graph merge, the literal analogue of merge_thread_graphs): every GPU
  builds the complete partial graph of its shard, occurrence rows and partial edges are exchanged by
  hash range, owners re-sort and sum the partial edge weights (an assembly lives in exactly one shard,
  build_internals.cpp:283-285).  Kept as a cross-check; it sorts every occurrence twice.

The concatenation of the slices in rank order is bit-identical to the single-GPU result (shard-count
invariance, the property the reference tests as thread-count invariance, tests/smoke/test_graph.py:67-127).

The collective choreography is independent of where the per-rank compute runs: ``engine`` supplies
local build / rows / splits / merge.  The product engine is :class:`HipEngine`; the CPU test-suite
drives the same choreography over gloo with a numpy engine defined in tests/.
"""
from __future__ import annotations

import ctypes
import math
import os
import time
from dataclasses import dataclass

import numpy as np


# SEQWIN_DIST_FORCE_COLLECTIVES=1 issues every collective even at world size 1 (exercises the RCCL calls on one GPU)
_FORCE_COLLECTIVES = os.environ.get("SEQWIN_DIST_FORCE_COLLECTIVES") == "1"


def partition_assemblies(n_assemblies: int, n_workers: int) -> list[tuple[int, int]]:
    """Contiguous ranges, first ``rem`` workers get one extra (cpp/src/seqwin/build.cpp:350-356)."""
    n_workers = max(1, n_workers)
    base, rem = divmod(n_assemblies, n_workers)
    out = []
    for t in range(n_workers):
        start = t * base + min(t, rem)
        out.append((start, start + base + (1 if t < rem else 0)))
    return out


def hash_bounds(n_parts: int) -> tuple[list[int], list[int]]:
    """Order-preserving splitters in [0, 2^64]: (node_bounds, edge_bounds), n_parts - 1 values each.

    Node hashes are uniform, so their splitters are j * 2^64 / P.  An edge is keyed by
    first = min(u, v), whose CDF is 1 - (1 - x)^2, so its splitters are the quantiles
    x_j = 1 - sqrt(1 - j / P)."""
    nb, eb = [], []
    for j in range(1, n_parts):
        nb.append(-((-j << 64) // n_parts))                                   # ceil(j * 2^64 / P)
        eb.append(min((1 << 64) - 1, (1 << 64) - math.isqrt(((n_parts - j) << 128) // n_parts)))
    return nb, eb


@dataclass
class Shard:
    """The assemblies one rank owns: a device batch plus its place in the job."""
    batch: object
    first_assembly: int
    n_assemblies_total: int
    _offs: object = None
    _global_offsets: object = None


class HipEngine:
    """Per-rank compute on the MI355X through the C ABI (include/seqwin_hip.h).

    ``staging="device"`` (default) exchanges device buffers directly (backend "nccl" = RCCL over xGMI).
    ``staging="host"`` stages the exchanged rows through host memory so that the same choreography can run
    over gloo (used to test the multi-process path on a box with a single GPU)."""

    def __init__(self, staging: str = "device"):
        import torch
        self.torch = torch
        self.gpu = torch.device("cuda", torch.cuda.current_device())
        self.device = self.gpu if staging == "device" else torch.device("cpu")

    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream().cuda_stream)

    def local_index(self, shard: Shard, k: int, w: int):
        return shard.batch.build_index(k, w, None, stream=self._stream())

    def record_offsets(self, shard: Shard) -> np.ndarray:
        if getattr(shard, "_offs", None) is None:   # ids are not needed here; fetch the offsets once
            shard._offs = shard.batch.record_offsets()
        return shard._offs

    def sizes(self, ix):
        return ix.sizes()

    def timings(self, ix) -> dict:
        return ix.timings()

    def splits(self, ix, node_bounds, edge_bounds):
        import ctypes

        from ._lib import c_u64, c_vp, check, lib
        n = len(node_bounds)
        nb = (c_u64 * max(n, 1))(*node_bounds)
        eb = (c_u64 * max(n, 1))(*edge_bounds)
        os_ = (c_u64 * max(n, 1))()
        es_ = (c_u64 * max(n, 1))()
        check(lib.sw_index_splits(ix._h, nb, eb, c_u64(n), os_, es_, c_vp(self._stream())))
        return list(os_)[:n], list(es_)[:n]

    def occ_rows(self, ix, rec_offset: int):
        from ._lib import c_u64, c_vp, check, lib
        n = ix.sizes()[0]
        rows = self.torch.empty((n, 2), dtype=self.torch.int64, device=self.gpu)
        check(lib.sw_index_occ_rows(ix._h, c_u64(rec_offset), c_vp(rows.data_ptr()), c_vp(self._stream())))
        return rows.to(self.device)

    def edge_rows(self, ix):
        from ._lib import c_vp, check, lib
        m = ix.sizes()[2]
        rows = self.torch.empty((m, 3), dtype=self.torch.int64, device=self.gpu)
        check(lib.sw_index_edge_rows(ix._h, c_vp(rows.data_ptr()), c_vp(self._stream())))
        return rows.to(self.device)

    def merge(self, occ_rows, edge_rows, kmer_base: int, record_offsets: np.ndarray, is_targets):
        import ctypes

        from ._lib import c_u64, c_vp, check, lib
        from .device import Index
        offs = np.ascontiguousarray(record_offsets, np.uint32)
        if is_targets is None:
            tar, na = None, len(offs) - 1
        else:
            t = np.ascontiguousarray(np.asarray(is_targets, np.bool_)).view(np.uint8)
            tar, na = t.ctypes.data_as(c_vp), len(t)
        occ_rows = occ_rows.to(self.gpu).contiguous()
        edge_rows = edge_rows.to(self.gpu).contiguous()
        # (no host synchronisation: a collective's result is ordered before later work on the current stream by
        #  torch.distributed, and the library launches on that same stream)
        h = c_vp()
        check(lib.sw_index_merge(c_vp(occ_rows.data_ptr()), c_u64(occ_rows.shape[0]), c_vp(edge_rows.data_ptr()),
                                 c_u64(edge_rows.shape[0]), c_u64(kmer_base), offs.ctypes.data_as(c_vp), tar, c_u64(na),
                                 c_vp(self._stream()), ctypes.byref(h)))
        return Index(h)

    # ---- tuple-exchange form ------------------------------------------------------------------------
    def sketch(self, shard: Shard, k: int, w: int):
        import ctypes

        from ._lib import c_u64, c_vp, check, lib
        h = c_vp()
        check(lib.sw_occ_sketch(shard.batch._h, c_u64(k), c_u64(w), c_vp(self._stream()), ctypes.byref(h)))
        n, ms = c_u64(), ctypes.c_double()
        check(lib.sw_occ_size(h, ctypes.byref(n), ctypes.byref(ms)))
        return _Occ(h, n.value, ms.value, shard.batch)

    def partition(self, occ, bounds, rec_offset: int):
        from ._lib import c_u64, c_vp, check, lib
        t = self.torch
        rows = t.empty((occ.n, 2), dtype=t.int64, device=self.gpu)
        nb = len(bounds)
        b = (c_u64 * max(nb, 1))(*bounds)
        cnt = (c_u64 * (nb + 1))()
        # no permutation array: the partition is stable, the occ handle keeps its offsets and sw_occ_adjacency walks them
        # again to read the returned ranks back in stream order
        check(lib.sw_occ_partition(occ._h, b, c_u64(nb), c_u64(rec_offset), c_vp(rows.data_ptr()), None, cnt, c_vp(self._stream())))
        return rows.to(self.device), None, [int(x) for x in cnt]

    def slice_build(self, rows, kmer_base: int, record_offsets: np.ndarray, is_targets):
        import ctypes

        from ._lib import c_u64, c_vp, check, lib
        from .device import Index
        t = self.torch
        rows = rows.to(self.gpu).contiguous()
        ranks = t.empty((rows.shape[0],), dtype=t.int32, device=self.gpu)
        offs = np.ascontiguousarray(record_offsets, np.uint32)
        if is_targets is None:
            tar, na = None, len(offs) - 1
        else:
            tt = np.ascontiguousarray(np.asarray(is_targets, np.bool_)).view(np.uint8)
            tar, na = tt.ctypes.data_as(c_vp), len(tt)
        h = c_vp()
        check(lib.sw_slice_build(c_vp(rows.data_ptr()), c_u64(rows.shape[0]), c_u64(kmer_base), offs.ctypes.data_as(c_vp),
                                 tar, c_u64(na), c_vp(ranks.data_ptr()), c_vp(self._stream()), ctypes.byref(h)))
        return Index(h), ranks.to(self.device)

    def node_hashes(self, ix):
        from ._lib import c_vp, check, lib
        t = self.torch
        out = t.empty((ix.sizes()[1],), dtype=t.int64, device=self.gpu)
        check(lib.sw_index_node_hashes(ix._h, c_vp(out.data_ptr()), c_vp(self._stream())))
        return out.to(self.device)

    def adjacency(self, occ, perm, ranks_by_row, n_bits: int, asm_bits: int, asm_base: int, rank_bounds):
        from ._lib import c_u64, c_vp, check, lib
        t = self.torch
        ranks_by_row = ranks_by_row.to(self.gpu).contiguous()
        m = max(occ.n - 1, 0)
        rows = t.empty((m,) if asm_bits else (m, 2), dtype=t.int64, device=self.gpu)
        nb = len(rank_bounds)
        b = (c_u64 * max(nb, 1))(*rank_bounds)
        cnt = (c_u64 * (nb + 1))()
        check(lib.sw_occ_adjacency(occ._h, c_vp(perm.data_ptr()) if perm is not None else None, c_vp(ranks_by_row.data_ptr()), c_u64(n_bits),
                                   c_u64(asm_bits), c_u64(asm_base), b, c_u64(nb), c_vp(rows.data_ptr()), cnt, c_vp(self._stream())))
        counts = [int(x) for x in cnt]
        return rows[:sum(counts)].to(self.device), counts

    def slice_edges(self, ix, adj_rows, n_bits: int, asm_bits: int, rank_hash) -> None:
        from ._lib import c_u64, c_vp, check, lib
        adj_rows = adj_rows.to(self.gpu).contiguous()
        rank_hash = rank_hash.to(self.gpu).contiguous()
        check(lib.sw_slice_edges(ix._h, c_vp(adj_rows.data_ptr()), c_u64(adj_rows.shape[0]), c_u64(n_bits),
                                 c_u64(asm_bits), c_vp(rank_hash.data_ptr()), c_vp(self._stream())))

    # ---- pairs form of the adjacency exchange (include/seqwin_hip.h: sw_occ_adjacency_pairs) ---------------------
    def ranks_marked(self, ix) -> bool:
        import ctypes

        from ._lib import check, lib
        m = ctypes.c_int()
        check(lib.sw_index_ranks_marked(ix._h, ctypes.byref(m)))
        return bool(m.value)

    def adjacency_pairs(self, occ, ranks_by_row, node_base, asm_base: int, rank_bounds):
        """ranks_by_row: slice-LOCAL ranks (uint32 patterns, repeat mark in bit 31) in partitioned-row order; node_base: prefix
        of the node counts of the tuple owners.  -> (keys by edge owner, counts, candidate rows, candidate counts, (lo_bits, hi_bits))"""
        from ._lib import c_u64, c_vp, check, lib
        t = self.torch
        ranks_by_row = ranks_by_row.to(self.gpu).contiguous()
        m = max(occ.n - 1, 0)
        keys = t.empty((m,), dtype=t.int64, device=self.gpu)
        nb = len(rank_bounds)
        b = (c_u64 * max(nb, 1))(*rank_bounds)
        base = (c_u64 * len(node_base))(*node_base)
        cnt, ccnt, bits = (c_u64 * (nb + 1))(), (c_u64 * (nb + 1))(), (c_u64 * 2)()
        check(lib.sw_occ_adjacency_pairs(occ._h, c_vp(ranks_by_row.data_ptr()), base, c_u64(len(node_base) - 1), c_u64(asm_base), b,
                                         c_u64(nb), c_vp(keys.data_ptr()), cnt, ccnt, bits, c_vp(self._stream())))
        counts, cand_counts = [int(x) for x in cnt], [int(x) for x in ccnt]
        cand = t.empty((sum(cand_counts), 2), dtype=t.int64, device=self.gpu)
        check(lib.sw_occ_candidates(occ._h, c_vp(cand.data_ptr()), c_vp(self._stream())))
        return keys[:sum(counts)].to(self.device), counts, cand.to(self.device), cand_counts, (int(bits[0]), int(bits[1]))

    def slice_edges_pairs(self, ix, keys, cand, key_bits, lo_base: int, asm_bits: int, rank_hash, node_base, pad: int) -> None:
        """rank_hash: the job-wide table, owner o's node hashes at [o * pad, o * pad + its count).
        ``keys`` is sorted where it lies (one half of the sort's double buffer): its contents are not preserved."""
        from ._lib import c_u64, c_vp, check, lib
        keys = keys.to(self.gpu).contiguous()
        cand = cand.to(self.gpu).contiguous()
        table = c_vp(0)                                  # None: the edges keep global ranks (edge_hash_requests / _attach follow)
        if rank_hash is not None:
            rank_hash = rank_hash.to(self.gpu).contiguous()
            table = c_vp(rank_hash.data_ptr())
        base = (c_u64 * len(node_base))(*node_base)
        check(lib.sw_slice_edges_pairs(ix._h, c_vp(keys.data_ptr()), c_u64(keys.shape[0]), c_vp(cand.data_ptr()),
                                       c_u64(cand.shape[0]), c_u64(key_bits[0]), c_u64(key_bits[1]), c_u64(lo_base), c_u64(asm_bits),
                                       table, base, c_u64(len(node_base) - 1), c_u64(pad), c_vp(self._stream())))

    # rank -> hash by request (no job-wide table): see sw_index_edge_hash_requests in include/seqwin_hip.h
    def edge_hash_requests(self, ix, node_base):
        """-> (owner-local ranks of the distinct endpoints of ix's edges, u32-in-int32 tensor in ascending global order;
        how many of them go to each node owner)"""
        import torch
        from ._lib import c_u64, c_vp, check, lib
        n_owners = len(node_base) - 1
        base = (c_u64 * len(node_base))(*node_base)
        cnt = (c_u64 * n_owners)()
        n = c_u64()
        check(lib.sw_index_edge_hash_requests(ix._h, base, c_u64(n_owners), cnt, ctypes.byref(n), c_vp(self._stream())))
        req = torch.empty((n.value,), dtype=torch.int32, device=self.gpu)
        check(lib.sw_index_edge_hash_request_rows(ix._h, c_vp(req.data_ptr()), c_vp(self._stream())))
        return req.to(self.device), [int(c) for c in cnt]

    def node_hash_lookup(self, ix, local_ranks):
        import torch
        from ._lib import c_u64, c_vp, check, lib
        r = local_ranks.to(self.gpu).contiguous()
        out = torch.empty((r.shape[0],), dtype=torch.int64, device=self.gpu)
        check(lib.sw_index_node_hash_lookup(ix._h, c_vp(r.data_ptr()), c_u64(r.shape[0]), c_vp(out.data_ptr()), c_vp(self._stream())))
        return out.to(self.device)

    def edge_hash_attach(self, ix, replies) -> None:
        from ._lib import c_u64, c_vp, check, lib
        r = replies.to(self.gpu).contiguous()
        check(lib.sw_index_edge_hash_attach(ix._h, c_vp(r.data_ptr()), c_u64(r.shape[0]), c_vp(self._stream())))

    def node_hash_part(self, ix, pad: int):
        """This slice's share of the job-wide rank -> hash table: its node hashes at the front of `pad` words (the rest
        is never read: sw_slice_edges_pairs goes through node_base)."""
        from ._lib import c_vp, check, lib
        out = self.torch.empty((pad,), dtype=self.torch.int64, device=self.gpu)
        check(lib.sw_index_node_hashes(ix._h, c_vp(out.data_ptr()), c_vp(self._stream())))
        return out.to(self.device)

    def free_occ(self, occ) -> None:
        occ.close()

    def close_index(self, ix) -> None:
        ix.close()

    def export(self, ix):
        return ix.export()

    def checksums(self, ix, kmer_base: int = 0, node_base: int = 0, edge_base: int = 0):
        return ix.checksums(kmer_base, node_base, edge_base)


class _Occ:
    """Handle of a device-resident ordered tuple stream (sw_occ)."""

    def __init__(self, h, n, sketch_ms, batch=None):
        self._h, self.n, self.sketch_ms = h, n, sketch_ms
        self._batch = batch   # sw_occ refers to its batch's record table (sw_occ_adjacency): keep the batch alive

    def close(self):
        from ._lib import lib
        if self._h:
            lib.sw_occ_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PhaseClock:
    """Phase boundaries of one sharded build.  With the HIP engine they are events on the current stream (no host
    synchronisation while the build is issued; the elapsed times are read after the caller's fence), otherwise host clocks."""

    def __init__(self, engine):
        t = getattr(engine, "torch", None)
        self._cuda = t.cuda if t is not None and getattr(engine, "gpu", None) is not None else None
        self.marks = []
        self.mark("start")

    def mark(self, name: str) -> None:
        if self._cuda is not None:
            e = self._cuda.Event(enable_timing=True)
            e.record()
            self.marks.append((name, e))
        else:
            self.marks.append((name, time.perf_counter()))

    def phases_ms(self) -> dict:
        """{phase: ms between the previous mark and the phase's own}; waits for the last mark."""
        out = {}
        if self._cuda is not None:
            self.marks[-1][1].synchronize()
        for (_, a), (name, b) in zip(self.marks, self.marks[1:]):
            ms = a.elapsed_time(b) if self._cuda is not None else (b - a) * 1e3
            out[name] = out.get(name, 0.0) + ms
        return out


class ShardedIndex:
    """This rank's slice (a hash range) of the merged graph, plus the job-wide metadata."""

    def __init__(self, engine, merged, record_offsets, timings, kmer_base, group=None, clock=None, info=None):
        self.engine, self.merged, self.group = engine, merged, group
        self.record_offsets, self._timings, self.kmer_base = record_offsets, timings, kmer_base
        self._clock, self.info = clock, dict(info or {})

    def sizes(self):
        return self.engine.sizes(self.merged)

    def timings(self) -> dict:
        return dict(self._timings)

    def phases_ms(self) -> dict:
        """Device time of the build's phases on this rank (sketch / partition / tuple exchange / slice build / ranks back /
        adjacency / key exchange / slice edges / hash requests); call after a synchronisation point."""
        return self._clock.phases_ms() if self._clock is not None else {}

    def export(self):
        return self.engine.export(self.merged)

    def _all_sizes(self):
        """sizes of every rank's slice: [(n_kmers, n_nodes, n_edges)] * world (one small all_gather)."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return [self.sizes()], 0
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        mine = torch.tensor(self.sizes(), dtype=torch.int64, device=self.engine.device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=self.group)
        return [tuple(int(v) for v in row) for row in torch.stack(parts).tolist()], rank

    def global_checksums(self):
        """Checksums of the concatenated arrays of all slices, identical on every rank: each rank checksums its slice
        at its offsets in the whole (sw_index_checksums_at), the shares add up modulo 2^64 (one all_reduce of 3 words)."""
        import torch
        import torch.distributed as dist
        sizes, rank = self._all_sizes()
        bases = [sum(s[j] for s in sizes[:rank]) for j in range(3)]
        mine = np.array(self.engine.checksums(self.merged, *bases), np.uint64)
        if len(sizes) == 1:
            return tuple(int(v) for v in mine)
        # sum modulo 2^64 without relying on signed overflow in the backend: 32-bit halves, each sum below 2^63
        halves = np.stack([mine & np.uint64(0xFFFFFFFF), mine >> np.uint64(32)]).astype(np.int64)
        t = torch.from_numpy(halves).to(self.engine.device)
        dist.all_reduce(t, group=self.group)
        lo, hi = t.cpu().numpy().astype(np.uint64)
        return tuple(int(v) for v in (lo + (hi << np.uint64(32))))

    def gather(self, dst: int = 0, group=None):
        """Concatenate all slices on rank ``dst`` -> (kmers, nodes, edges, record_offsets) or None elsewhere.

        The slices travel as raw bytes: sizes first (one all_gather), then each array as one ``dist.gather`` of uint8
        tensors padded to the largest slice -- no pickling, nothing is sent to ranks other than ``dst``."""
        import torch
        import torch.distributed as dist

        from ._core import EDGE_DTYPE, KMER_DTYPE, NODE_DTYPE
        group = group if group is not None else self.group
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return (*self.export(), self.record_offsets)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        sizes, _ = self._all_sizes()
        arrays = self.export()
        out = []
        for j, (arr, dt) in enumerate(zip(arrays, (KMER_DTYPE, NODE_DTYPE, EDGE_DTYPE))):
            pad = max(s[j] for s in sizes) * dt.itemsize
            buf = torch.zeros((max(pad, 1),), dtype=torch.uint8)
            raw = np.frombuffer(arr.tobytes(), np.uint8) if len(arr) else np.zeros(0, np.uint8)
            buf[:len(raw)] = torch.from_numpy(raw.copy())
            buf = buf.to(self.engine.device)
            parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
            # (dst is a rank of `group`; dist.gather addresses processes by their global rank)
            dist.gather(buf, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
            if rank == dst:
                out.append(np.concatenate([p.cpu().numpy()[:sizes[r][j] * dt.itemsize].view(dt) for r, p in enumerate(parts)]))
        if rank != dst:
            return None
        return (out[0], out[1], out[2], self.record_offsets)


def build_sharded_index_merge(shard: Shard, k: int, w: int, is_targets, engine=None, group=None) -> ShardedIndex:
    """Graph-merge form: build this rank's partial graph, exchange by hash range, finish this rank's slice.

    ``is_targets`` is the job-wide flag vector (one per assembly, all ranks pass the same) or None."""
    import torch
    import torch.distributed as dist

    engine = engine or HipEngine()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = engine.device
    t0 = time.perf_counter()

    ix = engine.local_index(shard, k, w)
    tm = {key: v for key, v in engine.timings(ix).items()}
    n_occ_local, _, n_edges_local = engine.sizes(ix)
    local_offs = np.asarray(engine.record_offsets(shard), np.uint32)

    # C0: record-count prefix (build_internals.cpp:334-355)
    if world > 1:
        all_offs = [None] * world
        dist.all_gather_object(all_offs, local_offs, group=group)
    else:
        all_offs = [local_offs]
    rec_base, glob = [], [np.zeros(1, np.uint32)]
    total = 0
    for o in all_offs:
        rec_base.append(total)
        glob.append((o[1:].astype(np.uint64) + total).astype(np.uint32))
        total += int(o[-1])
        if total > 0xFFFFFFFF:
            raise RuntimeError("Total number of FASTA records exceeds uint32 range")
    record_offsets = np.concatenate(glob)

    occ = engine.occ_rows(ix, rec_base[rank])
    edges = engine.edge_rows(ix)
    t1 = time.perf_counter()
    if world > 1:
        # C1: all-to-all-v of occurrence rows by hash range and of edge rows by `first` range
        nb, eb = hash_bounds(world)
        osp, esp = engine.splits(ix, nb, eb)
        ocut = [0] + [int(x) for x in osp] + [n_occ_local]
        ecut = [0] + [int(x) for x in esp] + [n_edges_local]
        send = torch.tensor([[ocut[j + 1] - ocut[j], ecut[j + 1] - ecut[j]] for j in range(world)], dtype=torch.int64, device=dev)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
        send_l, recv_l = send.tolist(), recv.tolist()
        r_occ = torch.empty((sum(r[0] for r in recv_l), 2), dtype=torch.int64, device=dev)
        r_edges = torch.empty((sum(r[1] for r in recv_l), 3), dtype=torch.int64, device=dev)
        dist.all_to_all_single(r_occ, occ, [r[0] for r in recv_l], [s[0] for s in send_l], group=group)
        dist.all_to_all_single(r_edges, edges, [r[1] for r in recv_l], [s[1] for s in send_l], group=group)
        owned = torch.tensor([r_occ.shape[0]], dtype=torch.int64, device=dev)
        owned_all = [torch.empty_like(owned) for _ in range(world)]
        dist.all_gather(owned_all, owned, group=group)
        kmer_base = int(sum(int(x.item()) for x in owned_all[:rank]))
    else:
        r_occ, r_edges, kmer_base = occ, edges, 0
    t2 = time.perf_counter()
    merged = engine.merge(r_occ, r_edges, kmer_base, record_offsets, is_targets)
    t3 = time.perf_counter()
    tm.update(n_occ_local=n_occ_local, local_build_wall_ms=(t1 - t0) * 1e3, exchange_wall_ms=(t2 - t1) * 1e3,
              merge_wall_ms=(t3 - t2) * 1e3)
    return ShardedIndex(engine, merged, record_offsets, tm, kmer_base, group)


def rank_bounds(n_parts: int, total_nodes: int) -> list[int]:
    """Order-preserving splitters of the node-rank space for edges keyed by rank_lo = min(rank_u, rank_v):
    the j/P quantile of min(u, v) for uniform u, v is 1 - sqrt(1 - j/P)."""
    return [total_nodes - math.isqrt(((n_parts - j) * total_nodes * total_nodes) // n_parts) for j in range(1, n_parts)]


def node_bases(node_counts) -> list[int]:
    """Prefix of the slice owners' node counts: global rank = node_bases[owner] + slice-local rank.
    SEQWIN_DIST_NODE_SPACING=K (tests) leaves K unused ranks behind every owner's nodes, so that a small job walks through the
    rank widths of BASELINE configs[4] (5e9 distinct minimizers: 33-bit ranks) -- every consumer goes through the bases."""
    spacing = int(os.environ.get("SEQWIN_DIST_NODE_SPACING", "0"))
    out = [0]
    for c in node_counts:
        out.append(out[-1] + int(c) + spacing)
    return out


def hash_route(table_words: int) -> str:
    """"table": every GPU gets the whole rank -> hash table (all-gather, 8 B per node of the job); "requests": the edge owners
    ask the node owners for the hashes of their edges' distinct endpoints (12 B per endpoint, no table in HBM).
    ``table_words`` is the size of the table as it would be allocated: world * max(node count of a slice), i.e. padded to the
    largest slice (skewed owners or SEQWIN_DIST_NODE_SPACING make that more than the node total).
    SEQWIN_DIST_HASH_ROUTE forces one; else the table up to SEQWIN_DIST_TABLE_LIMIT_MB (default 4096)."""
    forced = os.environ.get("SEQWIN_DIST_HASH_ROUTE")
    if forced in ("table", "requests"):
        return forced
    limit = int(os.environ.get("SEQWIN_DIST_TABLE_LIMIT_MB", "4096")) << 20
    return "requests" if int(table_words) * 8 > limit else "table"


def adjacency_asm_bits(n_bits: int, n_assemblies_total: int) -> int:
    """Width of the assembly field of a packed adjacency key ((rank_lo << n_bits | rank_hi) << asm_bits | assembly),
    or 0 when the key does not fit 64 bits and rows travel as {key, assembly} pairs."""
    ab = max(1, int(n_assemblies_total).bit_length())
    if os.environ.get("SEQWIN_AMD_NO_PACKED_EDGES"):
        return 0
    return ab if 2 * n_bits + ab <= 64 else 0


# RCCL 2.26 (ROCm 7.0, this image) delivers only the first 568.5 MiB of a peer's message in all_to_all_single -- measured at
# world size 1 with 1.19 GB messages, tests/tools/rccl_self_check.py: everything behind byte 596 115 456 of the output is left
# untouched, no error.  Messages therefore travel in rounds of at most SEQWIN_DIST_MSG_LIMIT_MB (default 256) per peer.
_MSG_LIMIT = int(os.environ.get("SEQWIN_DIST_MSG_LIMIT_MB", "256")) << 20


def _all_to_all_rows(out, rows, recv_counts, send_counts, group, global_max: int) -> None:
    """all_to_all_single(out, rows, recv_counts, send_counts) in rounds that keep every per-peer message below _MSG_LIMIT.

    ``global_max`` is the largest entry of the job-wide (source x destination) count matrix of this exchange: the number of
    rounds follows from it alone, so every rank issues the same sequence of collectives whatever its own counts are (a rank
    with nothing left to send or receive in a round takes part with zero-length messages)."""
    import torch
    import torch.distributed as dist
    row_bytes = rows.element_size() * math.prod(rows.shape[1:])                 # from the shape: an empty send has rows too
    cap = max(1, _MSG_LIMIT // max(1, row_bytes))
    send_counts, recv_counts = [int(c) for c in send_counts], [int(c) for c in recv_counts]
    global_max = int(global_max)
    if max(send_counts + recv_counts + [0]) > global_max:
        raise RuntimeError("all-to-all counts exceed the job-wide maximum they were announced with")
    if global_max <= cap:
        dist.all_to_all_single(out, rows.contiguous(), recv_counts, send_counts, group=group)
        return
    world = len(send_counts)
    s_off = [sum(send_counts[:p]) for p in range(world)]
    r_off = [sum(recv_counts[:p]) for p in range(world)]
    for done in range(0, global_max, cap):
        s_n = [min(cap, max(0, c - done)) for c in send_counts]
        r_n = [min(cap, max(0, c - done)) for c in recv_counts]
        piece = torch.cat([rows[s_off[p] + done:s_off[p] + done + s_n[p]] for p in range(world)])
        got = torch.empty((sum(r_n),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
        dist.all_to_all_single(got, piece, r_n, s_n, group=group)
        at = 0
        for p in range(world):
            out[r_off[p] + done:r_off[p] + done + r_n[p]] = got[at:at + r_n[p]]
            at += r_n[p]


_checked_groups = set()


def check_collectives(dev, group=None) -> None:
    """Start-up self-check of the collectives the build relies on, once per process and group: one all_to_all_single whose
    per-peer messages are ONE ELEMENT LARGER than the round size the exchanges use (SEQWIN_DIST_MSG_LIMIT_MB; capped at
    SEQWIN_DIST_SELFCHECK_MB, default 64 at world size 1 and the full round size between several ranks), filled with a position-dependent pattern
    and verified element for element on the receiver, plus an all_gather_into_tensor of the same size.  RCCL 2.26 at world
    size 1 silently delivered only the first half of large messages (NOTES.md, round 3): a transport that does so for the
    sizes in use fails HERE, loudly, instead of producing a graph with edges missing.  SEQWIN_DIST_SELFCHECK=0 skips it."""
    import torch
    import torch.distributed as dist
    key = (id(group), str(dev))
    if key in _checked_groups or os.environ.get("SEQWIN_DIST_SELFCHECK", "1") == "0" or not dist.is_initialized():
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # (r05) between distinct GPUs -- never exercised on any box of rounds 1-5 -- the check runs at the FULL round size: a per-peer message
    # of the size the exchanges really send (256 MiB: ~6 GB of buffers at world 8, some tens of ms once per process); at world size 1,
    # where the full-size path is covered by tests/test_gpu_fullsize.py, and over gloo (host memory: the CPU suite), it stays at 64 MiB
    limit = min(_MSG_LIMIT, int(os.environ.get("SEQWIN_DIST_SELFCHECK_MB", "64" if (world == 1 or dist.get_backend(group) != "nccl") else str(max(1, _MSG_LIMIT >> 20)))) << 20)
    n = limit // 8 + 1                                              # int64 elements per peer
    idx = torch.arange(n, dtype=torch.int64, device=dev)
    send = torch.cat([idx * 1000003 + (rank * world + p) * 7919 for p in range(world)])
    got = torch.full_like(send, -1)   # (poisoned: a block of the caching allocator may still hold an earlier check's pattern)
    dist.all_to_all_single(got, send, [n] * world, [n] * world, group=group)
    bad = 0
    for p in range(world):
        bad += int((got[p * n:(p + 1) * n] != idx * 1000003 + (p * world + rank) * 7919).sum().item())
    mine = idx * 31 + rank
    table = torch.full((world * n,), -1, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(table, mine, group=group)
    for p in range(world):
        bad += int((table[p * n:(p + 1) * n] != idx * 31 + p).sum().item())
    flag = torch.tensor([bad], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, group=group)
    if int(flag.item()):
        raise RuntimeError(f"collective self-check failed: {int(flag.item())} elements of {n}-element per-peer messages "
                           f"({n * 8 >> 20} MiB) arrived wrong or not at all -- lower SEQWIN_DIST_MSG_LIMIT_MB")
    _checked_groups.add(key)


def _all_gather_parts(table, mine, pad: int, group, async_op: bool = False):
    """all_gather_into_tensor(table[world * pad], mine[pad]); in rounds through a staging buffer when a part exceeds _MSG_LIMIT."""
    import torch
    import torch.distributed as dist
    if pad * mine.element_size() <= _MSG_LIMIT:
        return dist.all_gather_into_tensor(table, mine, group=group, async_op=async_op)
    world = table.numel() // pad
    cap = _MSG_LIMIT // mine.element_size()
    view = table.view(world, pad)
    for at in range(0, pad, cap):
        n = min(cap, pad - at)
        got = torch.empty((world, n), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(got.view(-1), mine[at:at + n].contiguous(), group=group)
        view[:, at:at + n] = got
    return None


def _exchange_rows(rows, counts, dev, group):
    """all_to_all_single of rows grouped by destination.  One all_gather makes the whole (source x destination) count
    matrix known everywhere (instead of a count all_to_all plus further gathers of totals); returns
    (received rows, per-source counts, matrix)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    send = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=dev)
    parts = torch.empty((world, world), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(parts.view(-1), send, group=group)
    matrix = parts.tolist()                                   # matrix[src][dst]
    recv_l = [int(matrix[src][rank]) for src in range(world)]
    out = torch.empty((sum(recv_l),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=dev)
    _all_to_all_rows(out, rows, recv_l, counts, group, max(max(r) for r in matrix))
    return out, recv_l, matrix


def _exchange_rows_pair(rows_a, counts_a, rows_b, counts_b, dev, group):
    """Two row exchanges (adjacency keys and candidate rows) whose split sizes travel in ONE all_gather."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    send = torch.tensor([int(c) for c in counts_a] + [int(c) for c in counts_b], dtype=torch.int64, device=dev)
    parts = torch.empty((world, 2 * world), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(parts.view(-1), send, group=group)
    matrix = parts.tolist()                                   # matrix[src] = counts_a by destination, then counts_b
    outs = []
    for rows, counts, off in ((rows_a, counts_a, 0), (rows_b, counts_b, world)):
        recv_l = [int(matrix[src][off + rank]) for src in range(world)]
        out = torch.empty((sum(recv_l),) + tuple(rows.shape[1:]), dtype=rows.dtype, device=dev)
        _all_to_all_rows(out, rows, recv_l, counts, group, max(max(r[off:off + world]) for r in matrix))
        outs.append(out)
    return outs


def _gather_ints(values, dev, group) -> list[int]:
    """all_gather of a few ints per rank -> flat list in rank order."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in (values if isinstance(values, (list, tuple)) else [values])], dtype=torch.int64, device=dev)
    out = torch.empty((dist.get_world_size(group) * t.numel(),), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, t, group=group)
    return [int(x) for x in out.tolist()]   # one device-to-host copy


def build_sharded_index(shard: Shard, k: int, w: int, is_targets, engine=None, group=None) -> ShardedIndex:
    """Tuple-exchange form: sketch the shard, exchange tuples by hash range, build this rank's slice of
    nodes / kmers / counts, return ranks to the sources, exchange adjacency rows, build this rank's edges.

    ``is_targets`` is the job-wide flag vector (one per assembly, all ranks pass the same) or None."""
    import torch
    import torch.distributed as dist

    engine = engine or HipEngine()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = engine.device
    multi = world > 1 or (_FORCE_COLLECTIVES and dist.is_initialized())
    if multi:
        check_collectives(dev, group)
    t0 = time.perf_counter()
    clock = _PhaseClock(engine)

    occ = engine.sketch(shard, k, w)
    t1 = time.perf_counter()
    clock.mark("sketch")

    # C0: record-count prefix (build_internals.cpp:334-355); depends only on the shard, so it is cached on it
    cached = getattr(shard, "_global_offsets", None)
    if cached is None or cached[0] != world:
        local_offs = np.asarray(engine.record_offsets(shard), np.uint32)
        if multi:
            all_offs = [None] * world
            dist.all_gather_object(all_offs, local_offs, group=group)
        else:
            all_offs = [local_offs]
        rec_base, glob, total = [], [np.zeros(1, np.uint32)], 0
        for o in all_offs:
            rec_base.append(total)
            glob.append((o[1:].astype(np.uint64) + total).astype(np.uint32))
            total += int(o[-1])
            if total > 0xFFFFFFFF:
                raise RuntimeError("Total number of FASTA records exceeds uint32 range")
        cached = (world, rec_base, np.concatenate(glob))
        try:
            shard._global_offsets = cached
        except Exception:
            pass
    _, rec_base, record_offsets = cached

    # C1: tuples to the owner of their hash range
    nb, _ = hash_bounds(world)
    rows, perm, cnt = engine.partition(occ, nb, rec_base[rank])
    clock.mark("partition")
    if multi:
        r_rows, recv_cnt, matrix = _exchange_rows(rows, cnt, dev, group)
        kmer_base = sum(int(matrix[src][r]) for r in range(rank) for src in range(world))   # rows owned by lower ranks
        tuple_max = max(max(r) for r in matrix)                  # the ranks travel back along the same matrix, transposed
    else:
        r_rows, recv_cnt, kmer_base, tuple_max = rows, cnt, 0, 0
    t2 = time.perf_counter()
    clock.mark("tuple_exchange")
    ix, r_ranks = engine.slice_build(r_rows, kmer_base, record_offsets, is_targets)
    tm = dict(engine.timings(ix))
    n_nodes = engine.sizes(ix)[1]
    t3 = time.perf_counter()
    clock.mark("slice_build")

    # C2: node ranks back to the sources; C3: rank -> hash table everywhere
    # A slice build marks, in bit 31 of the slice-local ranks it returns, the occurrences whose node recurs in their assembly:
    # if every slice did, the adjacency travels in its pairs form -- one 64-bit key per record plus the few records that can
    # repeat a pair inside an assembly.  The ranks go back as they are (32 bits, slice-local); the sources re-base them inside
    # the library (global rank = node_base[owner] + local rank, possibly more than 32 bits) -- no arithmetic on them here.
    marked = bool(getattr(engine, "ranks_marked", lambda ix: False)(ix))
    if multi:
        info = _gather_ints([n_nodes, 1 if marked else 0], dev, group)          # [n_nodes, marked] of every rank
        node_cnt, all_marked = info[0::2], all(info[1::2])
    else:
        node_cnt, all_marked = [n_nodes], marked
    node_base = node_bases(node_cnt)
    total_nodes = node_base[-1]
    pairs = all_marked and not os.environ.get("SEQWIN_AMD_NO_PACKED_EDGES")
    rb = rank_bounds(world, total_nodes)
    info = {"form": "pairs" if pairs else "rows", "hash_route": "table", "total_nodes": int(total_nodes)}
    if pairs:
        asm_bits = max(1, int(shard.n_assemblies_total).bit_length())
        # How the edge owners get the hashes of their edges' endpoints: the whole rank -> hash table on every GPU (8 B per node
        # of the job: 0.64 GB at 15 000 genomes, 40 GB at configs[4] with k >= 19), or -- above SEQWIN_DIST_TABLE_LIMIT_MB,
        # default 4096 -- by asking the node owners for the distinct endpoints only (hash_route "requests", below).
        table, pad, hash_work = None, max(1, max(node_cnt)), None
        by_request = hash_route(world * pad) == "requests"
        info["hash_route"] = "requests" if by_request else "table"
        if multi:
            ranks_by_row = torch.empty((occ.n,), dtype=torch.int32, device=dev)
            _all_to_all_rows(ranks_by_row, r_ranks, cnt, recv_cnt, group, tuple_max)
            if not by_request:
                mine = engine.node_hash_part(ix, pad)                           # this slice's hashes at the front of `pad` words
                table = torch.empty((world * pad,), dtype=torch.int64, device=dev)
                hash_work = _all_gather_parts(table, mine, pad, group, async_op=True)   # overlaps the adjacency build below
        else:
            ranks_by_row = r_ranks
            if not by_request:
                table = engine.node_hash_part(ix, pad)
        clock.mark("ranks_back")
        adj, acnt, cand, ccnt, key_bits = engine.adjacency_pairs(occ, ranks_by_row, node_base, shard.first_assembly, rb)
        info["key_bits"] = [int(b) for b in key_bits]
        clock.mark("adjacency")
        if multi:
            # (waited for BEFORE the next collective: torch runs synchronous collectives on the current stream and this one on
            #  its own -- two kernels of one communicator at once corrupt each other's data, seen at world size 1, r03)
            if hash_work is not None:
                hash_work.wait()
            r_adj, r_cand = _exchange_rows_pair(adj, acnt, cand, ccnt, dev, group)
            if world == 1 and os.environ.get("SEQWIN_DIST_DEBUG"):   # what went through the collectives must come back unchanged
                print("[dist debug] ranks", torch.equal(ranks_by_row, r_ranks), "adj", torch.equal(r_adj, adj), tuple(r_adj.shape),
                      tuple(adj.shape), "cand", torch.equal(r_cand, cand), tuple(cand.shape), "table",
                      table is None or torch.equal(table[:n_nodes], engine.node_hash_part(ix, pad)[:n_nodes]), "counts", acnt, ccnt,
                      flush=True)
        else:
            r_adj, r_cand = adj, cand
        t4 = time.perf_counter()
        clock.mark("key_exchange")
        engine.slice_edges_pairs(ix, r_adj, r_cand, key_bits, rb[rank - 1] if rank else 0, asm_bits, table, node_base, pad)
        clock.mark("slice_edges")
        if by_request:
            # the edges hold global ranks: their distinct endpoints, as owner-local ranks grouped by node owner, go to the
            # node owners; the hashes come back in the same order
            req, req_cnt = engine.edge_hash_requests(ix, node_base)
            if multi:
                got, got_cnt, req_matrix = _exchange_rows(req, req_cnt, dev, group)
                answers = engine.node_hash_lookup(ix, got)
                replies = torch.empty((int(req.shape[0]),), dtype=torch.int64, device=dev)
                _all_to_all_rows(replies, answers, req_cnt, got_cnt, group, max(max(r) for r in req_matrix))
            else:
                replies = engine.node_hash_lookup(ix, req)
            engine.edge_hash_attach(ix, replies)
            clock.mark("hash_requests")
    else:
        # {pair, assembly} rows or packed (pair, assembly) keys on GLOBAL 32-bit ranks (test knob; slices without marks)
        REP = 0x80000000
        if total_nodes >= 0xFFFFFFFF:
            raise RuntimeError("more than 2^32-2 nodes need the pairs form of the adjacency (every slice must mark its ranks)")
        rr = r_ranks.to(torch.int64) & 0xFFFFFFFF                               # uint32 patterns held in int32: widen UNSIGNED
        rr = ((rr & (REP - 1)) if marked else rr) + node_base[rank]
        if multi:
            ranks_by_row = torch.empty((occ.n,), dtype=torch.int32, device=dev)
            _all_to_all_rows(ranks_by_row, rr.to(torch.int32), cnt, recv_cnt, group, tuple_max)
            parts = [torch.empty((max(1, max(node_cnt)),), dtype=torch.int64, device=dev) for _ in range(world)]
            dist.all_gather(parts, engine.node_hash_part(ix, max(1, max(node_cnt))), group=group)
            rank_hash = torch.cat([p[:c] for p, c in zip(parts, node_cnt)])
        else:
            ranks_by_row, rank_hash = rr.to(torch.int32), engine.node_hashes(ix)
        clock.mark("ranks_back")
        n_bits = max(1, (total_nodes).bit_length())     # total_nodes <= 2^n_bits - 1
        asm_bits = adjacency_asm_bits(n_bits, shard.n_assemblies_total)
        adj, acnt = engine.adjacency(occ, perm, ranks_by_row, n_bits, asm_bits, shard.first_assembly, rb)
        clock.mark("adjacency")
        r_adj = _exchange_rows(adj, acnt, dev, group)[0] if multi else adj
        t4 = time.perf_counter()
        clock.mark("key_exchange")
        engine.slice_edges(ix, r_adj, n_bits, asm_bits, rank_hash)
        clock.mark("slice_edges")
    tm.update(engine.timings(ix))
    t5 = time.perf_counter()
    tm.update(sketch_ms=occ.sketch_ms, n_occ_local=occ.n, sketch_wall_ms=(t1 - t0) * 1e3, tuple_exchange_wall_ms=(t2 - t1) * 1e3,
              slice_build_wall_ms=(t3 - t2) * 1e3, rank_adj_exchange_wall_ms=(t4 - t3) * 1e3, slice_edges_wall_ms=(t5 - t4) * 1e3)
    engine.free_occ(occ)
    return ShardedIndex(engine, ix, record_offsets, tm, kmer_base, group, clock=clock, info=info)


def count_nodes_allreduce(shard: Shard, k: int, w: int, is_targets, engine=None, group=None):
    """Count-only path (SURVEY 8e "C2"; the north_star's "per-GPU minimizer tables merged by a single RCCL reduce"): the
    per-minimizer target / non-target genome counts and the penalty of the WHOLE job on every rank, without exchanging any
    occurrence.  Every rank indexes its own shard; the ranks agree on one sorted dictionary of minimizer hashes (an all-gather
    of each rank's distinct hashes, 8 B per local node), scatter their local counts into a dense [n_nodes, 2] u32 vector over
    that dictionary, and ONE all_reduce(sum) merges them -- an assembly lives in exactly one shard, so counts add
    (cpp/src/seqwin/build_internals.cpp:283-285; counting rule and penalty: cpp/src/seqwin/filter.cpp:62-136).

    Returns (hash u64[n_nodes] ascending, n_tar u32, n_neg u32, penalty f64) as numpy arrays, identical on every rank and equal to
    the ``hash / n_tar / n_neg / penalty`` fields of the merged graph's nodes.  No kmers, no edges: for consumers that only rank
    minimizers (the full graph takes build_sharded_index)."""
    import torch
    import torch.distributed as dist

    engine = engine or HipEngine()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    multi = world > 1 or (_FORCE_COLLECTIVES and dist.is_initialized())
    dev = engine.device
    tar_all = np.ascontiguousarray(np.asarray(is_targets, np.bool_).ravel())
    if len(tar_all) != shard.n_assemblies_total:
        raise ValueError("len(is_targets) must equal the number of assemblies of the job")
    n_tar_total, n_neg_total = int(tar_all.sum()), int((~tar_all).sum())
    if n_tar_total == 0:
        raise ValueError("is_targets must contain at least one target assembly")          # filter.cpp:55-57
    if n_neg_total == 0:
        raise ValueError("is_targets must contain at least one non-target assembly")      # filter.cpp:58-60
    if multi:
        check_collectives(dev, group)

    # -- this shard: occurrences grouped by node (hash order), each with its record; counts per node and class ----------------
    ix = engine.local_index(shard, k, w)
    offs = np.asarray(engine.record_offsets(shard), np.int64)            # local record offsets of the shard's assemblies
    rows = engine.occ_rows(ix, 0).to(dev)                                # [n, 2]: hash, pos | record << 32, in node order
    n = int(rows.shape[0])
    M63 = -(1 << 63)
    if n:
        h = rows[:, 0]
        rec = (rows[:, 1] >> 32) & 0xFFFFFFFF
        rec_asm = torch.from_numpy(np.repeat(np.arange(len(offs) - 1, dtype=np.int64), np.diff(offs))).to(dev)
        asm = rec_asm[rec]                                               # local assembly of every occurrence
        tar_local = torch.from_numpy(tar_all[shard.first_assembly:shard.first_assembly + len(offs) - 1].copy()).to(dev)
        head = torch.ones(n, dtype=torch.bool, device=dev)
        head[1:] = h[1:] != h[:-1]                                       # first occurrence of its node
        first = head.clone()
        first[1:] |= asm[1:] != asm[:-1]                                 # ... of its assembly in its node (occurrences are record-ordered)
        is_t = tar_local[asm]
        node_of = torch.cumsum(head.to(torch.int64), 0) - 1
        n_local = int(node_of[-1].item()) + 1
        cnt = torch.zeros((n_local, 2), dtype=torch.int64, device=dev)
        cnt[:, 0].index_add_(0, node_of, (first & is_t).to(torch.int64))
        cnt[:, 1].index_add_(0, node_of, (first & ~is_t).to(torch.int64))
        hashes = h[head] ^ M63                                           # unsigned order as signed order
    else:
        n_local = 0
        cnt = torch.zeros((0, 2), dtype=torch.int64, device=dev)
        hashes = torch.zeros((0,), dtype=torch.int64, device=dev)
    del rows

    # -- one dictionary for all ranks: the sorted union of the ranks' distinct hashes -------------------------------------------
    if multi:
        sizes = _gather_ints(n_local, dev, group)
        pad = max(1, max(sizes))
        mine = torch.zeros((pad,), dtype=torch.int64, device=dev)
        mine[:n_local] = hashes
        table = torch.empty((world * pad,), dtype=torch.int64, device=dev)
        _all_gather_parts(table, mine, pad, group)
        parts = [table[r * pad:r * pad + sizes[r]] for r in range(world)]
        dictionary = torch.unique(torch.cat(parts))                      # (sorted)
    else:
        dictionary = hashes                                              # (already sorted and distinct)
    n_nodes = int(dictionary.shape[0])

    # -- the single reduce: dense counts over the dictionary ------------------------------------------------------------------------
    dense = torch.zeros((n_nodes, 2), dtype=torch.int32, device=dev)
    if n_local:
        at = torch.searchsorted(dictionary, hashes)
        dense[at] = cnt.to(torch.int32)
    if multi:
        dist.all_reduce(dense, group=group)
    n_tar = dense[:, 0].to(torch.float64)
    n_neg = dense[:, 1].to(torch.float64)
    # filter.cpp:132-134, every operation on its own (no fused multiply-add): bit-identical to the reference's f64 penalty
    ft = n_tar * (1.0 / n_tar_total)
    fn = n_neg * (1.0 / n_neg_total)
    a = 1.0 - ft
    a = a * a
    b = fn * fn
    penalty = torch.sqrt(a + b)
    engine.close_index(ix) if hasattr(engine, "close_index") else None
    return ((dictionary ^ M63).cpu().numpy().view(np.uint64), dense[:, 0].cpu().numpy().astype(np.uint32),
            dense[:, 1].cpu().numpy().astype(np.uint32), penalty.cpu().numpy())


def build_graph_distributed(assembly_paths, k: int, w: int, is_targets=None, n_cpu: int = 1, group=None):
    """FASTA paths -> merged graph with every rank ingesting its own shard (rank 0 returns the arrays)."""
    import torch.distributed as dist

    from .device import Batch
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    paths = [str(p) for p in assembly_paths]
    start, end = partition_assemblies(len(paths), world)[rank]
    shard = Shard(Batch.from_fasta(paths[start:end], n_cpu=n_cpu), start, len(paths))
    sharded = build_sharded_index(shard, k, w, is_targets, group=group)
    return sharded.gather(0, group=group)
