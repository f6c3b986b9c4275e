"""ctypes loader for libseqwin_hip.so (the C-ABI boundary, include/seqwin_hip.h).

There is no CPU fallback: if the shared library is missing this module raises ImportError, and if no
HIP device is usable every compute call raises RuntimeError (SW_ERR_DEVICE).
"""
from __future__ import annotations

import ctypes
import os
import sys
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SEQWIN_AMD_LIB", _HERE / "libseqwin_hip.so"))

SW_OK, SW_ERR_RUNTIME, SW_ERR_VALUE, SW_ERR_DEVICE = 0, 1, 2, 3

c_u64 = ctypes.c_uint64
c_vp = ctypes.c_void_p


class Timings(ctypes.Structure):
    _fields_ = [("total_ms", ctypes.c_double), ("sketch_ms", ctypes.c_double), ("order_ms", ctypes.c_double),
                ("nodes_ms", ctypes.c_double), ("counts_ms", ctypes.c_double), ("edges_ms", ctypes.c_double),
                ("sketch_launches", c_u64), ("n_tiles", c_u64), ("total_bp", c_u64), ("n_windows", c_u64),
                ("ovf_tiles", c_u64), ("plan_ms", ctypes.c_double), ("plan_cached", c_u64)]


def _load() -> ctypes.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C seqwin_amd/csrc` (there is no CPU fallback)")
    # When torch is in the process its bundled libamdhip64 must be the one HIP runtime: load it first so
    # that our DT_NEEDED libamdhip64.so.7 resolves to the already-loaded copy (same SONAME).
    if "torch" not in sys.modules and os.environ.get("SEQWIN_AMD_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = ctypes.CDLL(str(LIB_PATH), mode=ctypes.RTLD_GLOBAL)
    lib.sw_last_error.restype = ctypes.c_char_p
    lib.sw_version.restype = ctypes.c_char_p
    lib.sw_graph_free.restype = None
    lib.sw_batch_free.restype = None
    lib.sw_index_free.restype = None
    for name in ("sw_graph_free", "sw_batch_free", "sw_index_free"):
        getattr(lib, name).argtypes = [c_vp]
    return lib


lib = _load()


def check(rc: int) -> None:
    """Map a status code to the exception class pybind11 raises for the reference's C++ exception."""
    if rc == SW_OK:
        return
    msg = lib.sw_last_error().decode("utf-8", "replace")
    if rc == SW_ERR_VALUE:
        raise ValueError(msg)
    raise RuntimeError(msg)
