import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

# The suite drives the library through switches that force size-dependent paths on small inputs, inject faults, lower bounds ...
# Those test hooks are compiled OUT of the release library (csrc/common.hpp: SW_TEST_GETENV); the suite loads the TEST library --
# same sources, hooks live -- unless told otherwise.  SEQWIN_AMD_RELEASE_LIB=1 keeps the release library (tests/test_release_library.py
# runs the parity tests that need no hook on it); an explicit SEQWIN_AMD_LIB wins over both.
import os  # noqa: E402

_TEST_LIB = ROOT / "seqwin_amd" / "libseqwin_hip_test.so"
if "SEQWIN_AMD_LIB" not in os.environ and os.environ.get("SEQWIN_AMD_RELEASE_LIB") != "1":
    if not _TEST_LIB.exists():   # (a fresh checkout: the libraries are build products, __graft_entry__.build() makes both)
        import subprocess
        r = subprocess.run(["make", "-C", str(ROOT / "seqwin_amd" / "csrc"), "-j6", "all", "test"], capture_output=True, text=True)
        if r.returncode != 0 or not _TEST_LIB.exists():
            raise RuntimeError(f"{_TEST_LIB} is missing and `make -C seqwin_amd/csrc all test` failed:\n{r.stderr[-2000:]}")
    os.environ["SEQWIN_AMD_LIB"] = str(_TEST_LIB)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return GOLDEN


@pytest.fixture(scope="session")
def manifest() -> dict:
    return json.loads((GOLDEN / "manifest.json").read_text())


@pytest.fixture(scope="session")
def smoke_paths() -> list[Path]:
    s = GOLDEN / "smoke"
    return [s / "targets/target-1.fasta", s / "targets/target-2.fasta",
            s / "non-targets/non-target-1.fasta", s / "non-targets/non-target-2.fasta"]


def load_case(case: dict):
    z = np.load(GOLDEN / "vectors" / f"{case['name']}.npz")
    paths = [GOLDEN / p for p in case["paths"]]
    return paths, z


def assert_graph_equal(got, exp_npz, ids=None):
    kmers, nodes, edges, offs, rec_ids = got
    assert kmers.dtype == exp_npz["kmers"].dtype and np.array_equal(kmers, exp_npz["kmers"])
    assert nodes.dtype == exp_npz["nodes"].dtype and np.array_equal(nodes, exp_npz["nodes"])
    assert edges.dtype == exp_npz["edges"].dtype and np.array_equal(edges, exp_npz["edges"])
    assert offs.dtype == np.uint32 and np.array_equal(offs, exp_npz["record_offsets"])
    if ids is not None:
        assert [list(t) for t in rec_ids] == ids
