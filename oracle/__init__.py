"""CPU oracle for the minimizer-index path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``seqwin_amd``) never does.

* ``oracle.build / minimize / nthash / get_penalty / filter_kmers`` call the C restatement
  (``oracle/seqwin_oracle.c`` -> ``oracle/libseqwin_oracle.so``, built by ``make -C oracle oracle``).
* ``oracle.load_ref()`` returns the REAL reference extension (``seqwin.graph._core`` compiled from
  ``/root/reference/cpp`` by ``make -C oracle ref`` into ``oracle/_ref/``) or ``None`` when it has
  not been built.  It exposes the reference's own ``_build_native``, ``_get_penalty_native`` and
  ``_filter_kmers_native`` (``cpp/src/bindings/python_bindings.cpp:43-169``).
"""
from __future__ import annotations

import ctypes
import glob
import importlib.util
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# SEQWIN_ORACLE_LIB: another build of the same restatement (the sanitizer build, `make -C oracle asan`)
_SO = Path(os.environ.get("SEQWIN_ORACLE_LIB", _HERE / "libseqwin_oracle.so"))

# Same layouts as the reference's wire format (cpp/include/seqwin/graph.hpp:15-53,
# src/seqwin/graph/__init__.py:40-58).
KMER_DTYPE = np.dtype([("pos", np.uint32), ("record_idx", np.uint32)])
NODE_DTYPE = np.dtype([
    ("hash", np.uint64), ("start", np.uintp), ("stop", np.uintp),
    ("n_tar", np.uint32), ("n_neg", np.uint32), ("penalty", np.float64),
])
EDGE_DTYPE = np.dtype([("first", np.uint64), ("second", np.uint64), ("weight", np.uintp)])

_lib = None


def build_oracle_lib(force: bool = False) -> Path:
    """Compile the C restatement if needed (gcc + zlib; no reference sources involved)."""
    src = _HERE / "seqwin_oracle.c"
    if "SEQWIN_ORACLE_LIB" in os.environ:
        return _SO
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "oracle"], stdout=subprocess.DEVNULL)
    return _SO


def build_ref(reference_root: str = "/root/reference") -> Path | None:
    """Compile the real reference into oracle/_ref/ when its sources are present."""
    if not Path(reference_root, "cpp", "src", "seqwin", "build.cpp").exists():
        return None
    subprocess.check_call(["make", "-C", str(_HERE), "ref", f"REF={reference_root}"],
                          stdout=subprocess.DEVNULL)
    hits = glob.glob(str(_HERE / "_ref" / "_core*.so"))
    return Path(hits[0]) if hits else None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build_oracle_lib()
        L = ctypes.CDLL(str(_SO))
        L.so_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc == 0:
        return
    msg = lib().so_last_error().decode("utf-8", "replace")
    raise (ValueError if rc == 2 else RuntimeError)(msg)


def load_ref():
    """The compiled reference extension module, or None if oracle/_ref/ has not been built."""
    hits = glob.glob(str(_HERE / "_ref" / "_core*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location("_core", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _seq_bytes(seq) -> bytes:
    return seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)


def nthash(seq, k: int):
    """(min_hash, out_hash, pos) of every valid k-mer of ``seq`` (NtHash::roll order)."""
    s = _seq_bytes(seq)
    cap = max(len(s), 1)
    mh = np.empty(cap, np.uint64); oh = np.empty(cap, np.uint64); pos = np.empty(cap, np.uint64)
    n = ctypes.c_size_t(0)
    _check(lib().so_nthash(s, ctypes.c_size_t(len(s)), ctypes.c_uint64(k),
                           mh.ctypes.data_as(ctypes.c_void_p), oh.ctypes.data_as(ctypes.c_void_p),
                           pos.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap), ctypes.byref(n)))
    return mh[:n.value].copy(), oh[:n.value].copy(), pos[:n.value].copy()


def minimize(seq, k: int, w: int):
    """btllib::minimize_sequence: (min_hash, out_hash, pos) arrays."""
    s = _seq_bytes(seq)
    cap = max(len(s), 1)
    mh = np.empty(cap, np.uint64); oh = np.empty(cap, np.uint64); pos = np.empty(cap, np.uint64)
    n = ctypes.c_size_t(0)
    _check(lib().so_minimize(s, ctypes.c_size_t(len(s)), ctypes.c_uint64(k), ctypes.c_uint64(w),
                             mh.ctypes.data_as(ctypes.c_void_p), oh.ctypes.data_as(ctypes.c_void_p),
                             pos.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap), ctypes.byref(n)))
    return mh[:n.value].copy(), oh[:n.value].copy(), pos[:n.value].copy()


def build(paths, k: int, w: int):
    """seqwin::build -> (kmers, nodes, edges, record_offsets, ids_by_assembly); also sets build.total_bp."""
    paths = [os.fsencode(str(p)) for p in paths]
    arr = (ctypes.c_char_p * max(len(paths), 1))(*paths)
    g = ctypes.c_void_p()
    _check(lib().so_build(arr, ctypes.c_size_t(len(paths)), ctypes.c_uint64(k), ctypes.c_uint64(w),
                          ctypes.byref(g)))
    try:
        sz = [ctypes.c_uint64() for _ in range(6)]
        lib().so_graph_sizes(g, *[ctypes.byref(x) for x in sz])
        nk, nn, ne, na, nb, bp = (x.value for x in sz)
        kmers = np.empty(nk, KMER_DTYPE); nodes = np.empty(nn, NODE_DTYPE); edges = np.empty(ne, EDGE_DTYPE)
        offs = np.empty(na + 1, np.uint32); blob = ctypes.create_string_buffer(max(nb, 1))
        lib().so_graph_export(g, kmers.ctypes.data_as(ctypes.c_void_p), nodes.ctypes.data_as(ctypes.c_void_p),
                              edges.ctypes.data_as(ctypes.c_void_p), offs.ctypes.data_as(ctypes.c_void_p), blob)
    finally:
        lib().so_graph_free(g)
    names = blob.raw[:nb].split(b"\0")[:-1] if nb else []
    ids, it = [], iter(names)
    for a in range(na):
        ids.append(tuple(next(it).decode("utf-8", "replace") for _ in range(int(offs[a + 1] - offs[a]))))
    build.total_bp = bp
    return kmers, nodes, edges, offs, ids


def get_penalty(kmers, nodes, record_offsets, is_targets) -> None:
    """seqwin::get_penalty, in place on ``nodes``."""
    kmers = np.ascontiguousarray(kmers, KMER_DTYPE)
    assert nodes.dtype == NODE_DTYPE and nodes.flags.c_contiguous and nodes.flags.writeable
    offs = np.ascontiguousarray(record_offsets, np.uint32)
    tar = np.ascontiguousarray(np.asarray(is_targets, np.bool_).ravel()).view(np.uint8)
    _check(lib().so_get_penalty(kmers.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(kmers)),
                                nodes.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(nodes)),
                                offs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(offs)),
                                tar.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(tar))))


def filter_kmers(kmers, nodes, used_hashes):
    """seqwin::filter_kmers -> (kmers_new, nodes_new)."""
    kmers = np.ascontiguousarray(kmers, KMER_DTYPE)
    nodes = np.ascontiguousarray(nodes, NODE_DTYPE)
    used = np.array(sorted(int(h) for h in used_hashes), np.uint64)
    nk, nn = ctypes.c_uint64(), ctypes.c_uint64()
    args = (kmers.ctypes.data_as(ctypes.c_void_p), nodes.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_uint64(len(nodes)), used.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(used)))
    _check(lib().so_filter_kmers(*args, None, None, ctypes.byref(nk), ctypes.byref(nn)))
    ko = np.empty(nk.value, KMER_DTYPE); no = np.empty(nn.value, NODE_DTYPE)
    _check(lib().so_filter_kmers(*args, ko.ctypes.data_as(ctypes.c_void_p), no.ctypes.data_as(ctypes.c_void_p),
                                 ctypes.byref(nk), ctypes.byref(nn)))
    return ko, no


def read_fasta(path):
    """read_fasta (cpp/src/utils/fasta_reader.cpp:207-213): list of (id, sequence bytes).
    Sequences containing NUL bytes are not representable through this helper."""
    blob = ctypes.c_void_p(); n = ctypes.c_size_t(); nrec = ctypes.c_size_t()
    _check(lib().so_read_fasta(os.fsencode(str(path)), ctypes.byref(blob), ctypes.byref(n), ctypes.byref(nrec)))
    try:
        raw = ctypes.string_at(blob, n.value)
    finally:
        lib().so_free(blob)
    parts = raw.split(b"\0")[:-1] if n.value else []
    return [(parts[2 * i].decode("utf-8", "replace"), parts[2 * i + 1]) for i in range(nrec.value)]
