"""The RELEASE library (seqwin_amd/libseqwin_hip.so: test hooks compiled out) through the parity tests that need no hook.

The suite itself loads the test library (tests/conftest.py) because most of it drives size-dependent paths through switches the
release library does not read.  What ships is checked here: the smoke set, the reference's golden graph, the reference-generated
vectors, the differential fuzz against the oracle, synthetic and ragged batches, the f64 penalty, the resident route -- in a fresh
interpreter with SEQWIN_AMD_RELEASE_LIB=1 -- and a build under one of the hooks, which must change nothing."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def test_parity_tests_pass_on_the_release_library():
    env = {k: v for k, v in os.environ.items() if k != "SEQWIN_AMD_LIB"}
    env["SEQWIN_AMD_RELEASE_LIB"] = "1"
    sel = ("smoke or golden or fuzz_build or synthetic_batch or ragged or penalty_f64 or identity_test or reference_vectors or "
           "full_size_properties or low_complexity or multi_device_build_on_a_synthetic_job or pipelined_build or low_memory_is_honoured")
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_gpu_parity.py"), "-x", "-q", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], capture_output=True, text=True, cwd=str(ROOT), env=env, timeout=900)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail, tail
    n = int(tail.split(" passed")[0].split()[-1])
    assert n >= 12, tail


def test_release_library_ignores_a_test_hook():
    """SEQWIN_AMD_FAULT_INJECT=rank makes the TEST library mis-rank its radix passes (the order guards' test); the release library
    must not even look at it: no guard trips, same arrays."""
    code = (
        "import os, ctypes, numpy as np\n"
        "os.environ['SEQWIN_AMD_FAULT_INJECT'] = 'rank'; os.environ['SEQWIN_AMD_SORT'] = 'own'\n"
        "from seqwin_amd._lib import LIB_PATH, lib\n"
        "from seqwin_amd.device import Batch\n"
        "assert str(LIB_PATH).endswith('libseqwin_hip.so'), LIB_PATH\n"
        "b = Batch.synthetic(64, 10, 100000, n_ancestors=3, snp_ppm=10000, seed=5)\n"
        "ix = b.build_index(21, 200, [i % 2 == 0 for i in range(64)])\n"
        "a, c = ctypes.c_uint64(), ctypes.c_uint64()\n"
        "assert lib.sw_order_guard_trips(ctypes.byref(a), ctypes.byref(c)) == 0\n"
        "print('trips', a.value, c.value, ix.sizes())\n")
    env = {k: v for k, v in os.environ.items() if k != "SEQWIN_AMD_LIB"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(ROOT), env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "trips 0 0" in r.stdout, r.stdout
