"""Device-resident pipeline: batches of 2-bit packed assemblies in HBM and indexes built from them.

Thin object wrappers over the ``sw_batch_*`` / ``sw_index_*`` / ``sw_sketch`` entry points of
include/seqwin_hip.h.  This is what ``sw_build`` is made of; bench.py and the multi-GPU driver use it
directly so that the timed region starts with inputs already resident in HBM.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from ._core import EDGE_DTYPE, KMER_DTYPE, NODE_DTYPE, _ptr, _split_ids
from ._lib import Timings, c_u64, c_vp, check, lib


def device_count() -> int:
    return int(lib.sw_device_count())


def set_device(device: int) -> None:
    check(lib.sw_set_device(ctypes.c_int(device)))


def pool_trim() -> None:
    """Hand the library's cached, unused device blocks back to the driver (for a process that shares HBM with torch)."""
    lib.sw_pool_trim()


class Batch:
    """A set of assemblies, 2-bit packed and resident on the current device."""

    def __init__(self, handle: c_vp):
        self._h = handle

    @classmethod
    def from_fasta(cls, assembly_paths, n_cpu: int = 1) -> "Batch":
        paths = [os.fsencode(str(p)) for p in assembly_paths]
        arr = (ctypes.c_char_p * max(len(paths), 1))(*paths)
        h = c_vp()
        check(lib.sw_batch_from_fasta(arr, ctypes.c_size_t(len(paths)), c_u64(n_cpu), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def synthetic(cls, n_genomes: int, records_per_genome: int, record_len: int, n_ancestors: int = 1,
                  snp_ppm: int = 10000, seed: int = 1, first_genome: int = 0) -> "Batch":
        """Genomes [first_genome, first_genome + n_genomes) of the synthetic job ``seed``: a shard holds exactly the
        bases and record ids the unsharded batch holds for these genomes."""
        h = c_vp()
        check(lib.sw_batch_synthetic_shard(c_u64(n_genomes), c_u64(records_per_genome), c_u64(record_len),
                                           c_u64(n_ancestors), c_u64(snp_ppm), c_u64(seed), c_u64(first_genome),
                                           ctypes.byref(h)))
        return cls(h)

    @classmethod
    def synthetic_ragged(cls, n_genomes: int, genome_bp: int, n_ancestors: int = 1, snp_ppm: int = 10000, seed: int = 1,
                         first_genome: int = 0) -> "Batch":
        """Ragged draft assemblies (sw_batch_synthetic_ragged): 20-300 contigs of 200 bp ... 1.5 Mbp per genome, scaffold gaps of
        10-1000 N in one contig of ten."""
        h = c_vp()
        check(lib.sw_batch_synthetic_ragged(c_u64(n_genomes), c_u64(genome_bp), c_u64(n_ancestors), c_u64(snp_ppm), c_u64(seed),
                                            c_u64(first_genome), ctypes.byref(h)))
        return cls(h)

    def info(self) -> dict:
        v = [c_u64() for _ in range(4)]
        check(lib.sw_batch_info(self._h, *[ctypes.byref(x) for x in v]))
        return dict(n_assemblies=v[0].value, n_records=v[1].value, total_bp=v[2].value, device_bytes=v[3].value)

    def record_offsets(self) -> np.ndarray:
        offs = np.empty(self.info()["n_assemblies"] + 1, np.uint32)
        nb = c_u64()
        check(lib.sw_batch_records(self._h, _ptr(offs), None, c_u64(0), ctypes.byref(nb)))
        return offs

    def records(self):
        """(record_offsets, ids_by_assembly)"""
        na = self.info()["n_assemblies"]
        offs = np.empty(na + 1, np.uint32)
        nb = c_u64()
        check(lib.sw_batch_records(self._h, _ptr(offs), None, c_u64(0), ctypes.byref(nb)))
        blob = ctypes.create_string_buffer(max(nb.value, 1))
        check(lib.sw_batch_records(self._h, _ptr(offs), blob, c_u64(nb.value), ctypes.byref(nb)))
        return offs, _split_ids(blob.raw[:nb.value], offs)

    def record(self, record_idx: int) -> bytes:
        n = c_u64()
        check(lib.sw_batch_record(self._h, c_u64(record_idx), None, c_u64(0), ctypes.byref(n)))
        buf = ctypes.create_string_buffer(max(n.value, 1))
        check(lib.sw_batch_record(self._h, c_u64(record_idx), buf, c_u64(n.value), ctypes.byref(n)))
        return buf.raw[:n.value]

    def sketch(self, kmerlen: int, windowsize: int, stream: int = 0):
        """All minimizers in (record_idx, pos) order: (out_hash[u64], kmers[KMER_DTYPE])."""
        n = c_u64()
        check(lib.sw_sketch(self._h, c_u64(kmerlen), c_u64(windowsize), c_vp(stream), None, None, c_u64(0),
                            ctypes.byref(n)))
        oh = np.empty(n.value, np.uint64)
        km = np.empty(n.value, KMER_DTYPE)
        check(lib.sw_sketch(self._h, c_u64(kmerlen), c_u64(windowsize), c_vp(stream), _ptr(oh), _ptr(km),
                            c_u64(n.value), ctypes.byref(n)))
        return oh[:n.value], km[:n.value]

    def build_index(self, kmerlen: int, windowsize: int, is_targets=None, stream: int = 0) -> "Index":
        h = c_vp()
        if is_targets is None:
            tar, n = None, 0
        else:
            t = np.ascontiguousarray(np.asarray(is_targets, np.bool_)).view(np.uint8)
            tar, n = _ptr(t), len(t)
        check(lib.sw_index_build(self._h, c_u64(kmerlen), c_u64(windowsize), tar, c_u64(n), c_vp(stream),
                                 ctypes.byref(h)))
        return Index(h)

    def close(self) -> None:
        if self._h:
            lib.sw_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """kmers / nodes / edges of a batch, resident on the device."""

    def __init__(self, handle: c_vp):
        self._h = handle

    def sizes(self):
        v = [c_u64() for _ in range(3)]
        check(lib.sw_index_sizes(self._h, *[ctypes.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def timings(self) -> dict:
        t = Timings()
        check(lib.sw_index_timings(self._h, ctypes.byref(t)))
        return {name: getattr(t, name) for name, _ in Timings._fields_}

    def export(self):
        nk, nn, ne = self.sizes()
        kmers = np.empty(nk, KMER_DTYPE)
        nodes = np.empty(nn, NODE_DTYPE)
        edges = np.empty(ne, EDGE_DTYPE)
        check(lib.sw_index_export(self._h, _ptr(kmers), _ptr(nodes), _ptr(edges)))
        return kmers, nodes, edges

    def threshold_sums(self):
        """(sum n_tar, sum n_tar^2, sum n_tar*n_neg) over the nodes, computed on device (kmers.py:426-429)."""
        v = (c_u64 * 3)()
        check(lib.sw_index_threshold_sums(self._h, v))
        return tuple(int(x) for x in v)

    def filter_graph(self, edge_weight_th: float) -> "Index":
        """kmers._filter_edges_and_nodes on device: edges with weight > th and their endpoint nodes (no kmers)."""
        h = c_vp()
        check(lib.sw_index_filter_graph(self._h, c_u64(int(np.uintp(edge_weight_th))), ctypes.byref(h)))
        return Index(h)

    def filter_kmers(self, nodes_from: "Index", used_hashes) -> "Index":
        """filter_kmers on device: nodes of ``nodes_from`` whose hash is in ``used_hashes`` + their kmers from self."""
        used = np.fromiter((int(x) for x in used_hashes), dtype=np.uint64)
        h = c_vp()
        check(lib.sw_index_filter_kmers(self._h, nodes_from._h, _ptr(used), c_u64(len(used)), ctypes.byref(h)))
        return Index(h)

    def save_npz(self, path, record_offsets) -> None:
        """Write ``graph.npz`` exactly as ``--save-graph`` does (src/seqwin/core.py:134-145)."""
        kmers, nodes, edges = self.export()
        np.savez(path, allow_pickle=False, kmers=kmers, nodes=nodes, edges=edges,
                 record_offsets=np.asarray(record_offsets, np.uint32))

    def checksums(self, kmer_base: int = 0, node_base: int = 0, edge_base: int = 0):
        """(kmers, nodes, edges) checksums; with bases: this slice's share of the checksums of the concatenated arrays
        (the shares of all slices add up modulo 2^64)."""
        v = (c_u64 * 3)()
        check(lib.sw_index_checksums_at(self._h, c_u64(kmer_base), c_u64(node_base), c_u64(edge_base), v))
        return tuple(int(x) for x in v)

    def verify(self, n_assemblies: int, scored: bool = True) -> dict:
        """Device-side self-check (sw_index_verify): violation counts of the output's structural properties."""
        v = (c_u64 * 10)()
        check(lib.sw_index_verify(self._h, c_u64(n_assemblies), ctypes.c_int(1 if scored else 0), v))
        names = ("node_order", "node_ranges", "kmer_order", "edge_order", "edge_first_gt_second", "edge_weight_range",
                 "edge_endpoint_missing", "count_range", "weight_sum", "reserved")
        return dict(zip(names, (int(x) for x in v)))

    def close(self) -> None:
        if self._h:
            lib.sw_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_G = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
    x ^= x >> np.uint64(31)
    return x


CHECKSUM_SCHEME = 2   # r06: every field carries the element index (1: kmers, nodes.hash and edges.first only)
_K1, _K2, _K3, _K4 = (np.uint64(0xA0761D6478BD642F), np.uint64(0xE7037ED1A0B428DB), np.uint64(0x8EBC6AF09C88C6E3),
                      np.uint64(0x589965CC75374CC3))


def host_checksums(kmers, nodes, edges, kmer_base: int = 0, node_base: int = 0, edge_base: int = 0):
    """numpy restatement of sw_index_checksums / sw_index_checksums_at (csrc/device.hpp: ck_kmer / ck_node / ck_edge) for host
    arrays.  Every field of an element is mixed with the element's index (r06): kmers; nodes' hash, start, stop, (n_tar, n_neg) and
    the penalty's bit pattern; edges' first, second, weight -- whole rows, as tests/smoke/test_graph.py:281-291 compares them."""
    with np.errstate(over="ignore"):
        x = (np.arange(len(kmers), dtype=np.uint64) + np.uint64(kmer_base)) * _G
        a = _mix64(x + (kmers["pos"].astype(np.uint64) | (kmers["record_idx"].astype(np.uint64) << np.uint64(32)))).sum(dtype=np.uint64)
        x = (np.arange(len(nodes), dtype=np.uint64) + np.uint64(node_base)) * _G
        b = (_mix64(x + nodes["hash"]) + _mix64((x ^ _K1) + nodes["start"].astype(np.uint64)) +
             _mix64((x ^ _K2) + nodes["stop"].astype(np.uint64)) +
             _mix64((x ^ _K3) + ((nodes["n_tar"].astype(np.uint64) << np.uint64(32)) | nodes["n_neg"].astype(np.uint64))) +
             _mix64((x ^ _K4) + np.ascontiguousarray(nodes["penalty"]).view(np.uint64))).sum(dtype=np.uint64)
        x = (np.arange(len(edges), dtype=np.uint64) + np.uint64(edge_base)) * _G
        c = (_mix64(x + edges["first"]) + _mix64((x ^ _K1) + edges["second"].astype(np.uint64)) +
             _mix64((x ^ _K2) + edges["weight"].astype(np.uint64))).sum(dtype=np.uint64)
    return int(a), int(b), int(c)
