"""Drop-in for the reference's native module ``seqwin.graph._core``.

Same three callables, same argument meaning, same return layouts and error classes as the pybind11
module in the reference (cpp/src/bindings/python_bindings.cpp:43-169), implemented over the C ABI of
libseqwin_hip.so (include/seqwin_hip.h) with ctypes.  ctypes releases the GIL for the duration of every
foreign call, as the reference does with py::gil_scoped_release (:59,117,151).
"""
from __future__ import annotations

import ctypes
import gc
import logging
import operator
import os

import numpy as np

from ._lib import c_u64, c_vp, check, lib

# Wire formats: cpp/include/seqwin/graph.hpp:15-53 via PYBIND11_NUMPY_DTYPE (python_bindings.cpp:44-46)
KMER_DTYPE = np.dtype([("pos", np.uint32), ("record_idx", np.uint32)])
NODE_DTYPE = np.dtype([("hash", np.uint64), ("start", np.uintp), ("stop", np.uintp),
                       ("n_tar", np.uint32), ("n_neg", np.uint32), ("penalty", np.float64)])
EDGE_DTYPE = np.dtype([("first", np.uint64), ("second", np.uint64), ("weight", np.uintp)])


def _size_t(value, name: str) -> int:
    """pybind11's std::size_t caster: Python ints (and __index__), no floats, no negatives."""
    if isinstance(value, float):
        raise TypeError(f"{name}: expected an integer, got float")
    try:
        v = operator.index(value)
    except TypeError:
        raise TypeError(f"{name}: expected an integer, got {type(value).__name__}") from None
    if v < 0 or v >= 1 << 64:
        raise TypeError(f"{name}: {v} does not fit std::size_t")
    return v


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(c_vp)


def _noconvert(a, dtype: np.dtype, name: str) -> np.ndarray:
    """py::array_t<T, c_style> with .noconvert(): an ndarray of exactly this dtype, C-contiguous."""
    if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous:
        got = f"ndarray[{a.dtype}]" if isinstance(a, np.ndarray) else type(a).__name__
        raise TypeError(f"{name}: incompatible argument, expected C-contiguous numpy.ndarray[{dtype}], got {got}")
    return a


def _split_ids(blob: bytes, record_offsets: np.ndarray) -> list[tuple[str, ...]]:
    """ids_by_assembly of python_bindings.cpp:73-80 from the exported blob (every id followed by a NUL) and the record offsets.
    One decode and one split for the whole blob, and no cyclic-GC passes while the tuples are made (15 000 assemblies x 50 contigs:
    0.05 s instead of 0.2 s of a 1.7 s call); an id that is not UTF-8 raises UnicodeDecodeError, as pybind11's cast does."""
    names = blob.decode("utf-8").split("\0") if blob else []   # (one empty string behind the last NUL: no offset reaches it)
    offs = record_offsets.tolist()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        return [tuple(names[offs[a]:offs[a + 1]]) for a in range(len(offs) - 1)]
    finally:
        if was_enabled:
            gc.enable()


# Native log lines go to Python's root logger, as log_python does in the reference (cpp/src/utils/logging.cpp:9-29).
from ._abi import LOG_FN as _LOG_FN  # noqa: E402  (CFUNCTYPE(None, c_char_p, c_char_p): sw_log_fn)


def _forward_log(level, message):
    try:
        fn = getattr(logging.getLogger(), (level or b"info").decode("ascii", "replace"), None) or logging.getLogger().info
        fn((message or b"").decode("utf-8", "replace"))
    except Exception:   # never let a logging problem unwind through the C frame
        pass


_log_callback = _LOG_FN(_forward_log)       # kept alive for the lifetime of the module
lib.sw_set_log_callback(_log_callback)


def _build_native(assembly_paths, kmerlen, windowsize, n_cpu=1, low_memory=False):
    """seqwin::build on the GPU.  Returns (kmers, nodes, edges, record_offsets, ids_by_assembly)
    exactly as python_bindings.cpp:50-90 does."""
    if isinstance(assembly_paths, (str, bytes)) or not hasattr(assembly_paths, "__iter__"):
        raise TypeError("assembly_paths: expected a list of str")
    paths = []
    for p in assembly_paths:
        if not isinstance(p, (str, bytes)):
            raise TypeError("assembly_paths: expected a list of str")
        paths.append(os.fsencode(p))
    k = _size_t(kmerlen, "kmerlen")
    w = _size_t(windowsize, "windowsize")
    n_cpu = _size_t(n_cpu, "n_cpu")
    arr = (ctypes.c_char_p * max(len(paths), 1))(*paths)
    g = c_vp()
    check(lib.sw_build(arr, ctypes.c_size_t(len(paths)), c_u64(k), c_u64(w), c_u64(n_cpu),
                       ctypes.c_int(1 if low_memory else 0), ctypes.byref(g)))
    try:
        sz = [c_u64() for _ in range(6)]
        check(lib.sw_graph_sizes(g, *[ctypes.byref(x) for x in sz]))
        nk, nn, ne, na, nb, _bp = (x.value for x in sz)
        kmers = np.empty(nk, KMER_DTYPE)
        nodes = np.empty(nn, NODE_DTYPE)
        edges = np.empty(ne, EDGE_DTYPE)
        record_offsets = np.empty(na + 1, np.uint32)
        blob = ctypes.create_string_buffer(max(nb, 1))
        check(lib.sw_graph_export(g, _ptr(kmers), _ptr(nodes), _ptr(edges), _ptr(record_offsets), blob))
    finally:
        lib.sw_graph_free(g)
    return kmers, nodes, edges, record_offsets, _split_ids(blob.raw[:nb], record_offsets)


def _hashes_array(used_hashes) -> np.ndarray:
    """std::vector<uint64_t> from any iterable of ints, as pybind11's caster fills it (python_bindings.cpp:137-150).  In practice a
    frozenset[np.uint64] of up to millions of node hashes (kmers.py:312): NumPy converts a list of unsigned or non-negative integers
    in one go (0.15 s per million instead of 0.43 s element by element); anything else -- floats, negatives, objects -- goes through
    the per-element check, which raises what the caster raises."""
    items = list(used_hashes)
    if not items:
        return np.empty(0, np.uint64)
    try:
        a = np.asarray(items)
    except Exception:
        a = None
    if a is not None and a.ndim == 1 and a.shape[0] == len(items):
        if a.dtype == np.uint64:
            return np.ascontiguousarray(a)
        if a.dtype.kind == "i" and int(a.min()) >= 0:
            return a.astype(np.uint64)
    return np.fromiter((_size_t(h, "used_hashes") for h in items), dtype=np.uint64, count=len(items))


def _get_penalty_native(kmers, nodes, record_offsets, is_targets, n_cpu=1):
    """seqwin::get_penalty on the GPU, in place on ``nodes`` (python_bindings.cpp:92-135)."""
    kmers = _noconvert(kmers, KMER_DTYPE, "kmers")
    nodes = _noconvert(nodes, NODE_DTYPE, "nodes")
    record_offsets = _noconvert(record_offsets, np.dtype(np.uint32), "record_offsets")
    is_targets = _noconvert(is_targets, np.dtype(np.bool_), "is_targets")
    n_cpu = _size_t(n_cpu, "n_cpu")
    if not nodes.flags.writeable:
        raise ValueError("nodes must be writable")  # python_bindings.cpp:104-106
    # the reference reads shape[0] of each argument (:113-115)
    dim0 = lambda a: a.shape[0] if a.ndim else 1  # noqa: E731
    check(lib.sw_get_penalty(_ptr(kmers), c_u64(dim0(kmers)), _ptr(nodes), c_u64(dim0(nodes)),
                             _ptr(record_offsets), c_u64(dim0(record_offsets)),
                             _ptr(is_targets.view(np.uint8)), c_u64(dim0(is_targets)), c_u64(n_cpu)))
    return None


def _filter_kmers_native(kmers, nodes, used_hashes):
    """seqwin::filter_kmers on the GPU -> (kmers_new, nodes_new) (python_bindings.cpp:137-168)."""
    kmers = _noconvert(kmers, KMER_DTYPE, "kmers")
    nodes = _noconvert(nodes, NODE_DTYPE, "nodes")
    if isinstance(used_hashes, (str, bytes)):
        raise TypeError("used_hashes: expected an iterable of int")
    used = _hashes_array(used_hashes)
    nk, nn = c_u64(), c_u64()
    args = (_ptr(kmers), c_u64(len(kmers)), _ptr(nodes), c_u64(len(nodes)), _ptr(used), c_u64(len(used)))
    check(lib.sw_filter_kmers(*args, None, None, ctypes.byref(nk), ctypes.byref(nn)))
    kmers_new = np.empty(nk.value, KMER_DTYPE)
    nodes_new = np.empty(nn.value, NODE_DTYPE)
    check(lib.sw_filter_kmers(*args, _ptr(kmers_new), _ptr(nodes_new), ctypes.byref(nk), ctypes.byref(nn)))
    return kmers_new, nodes_new
