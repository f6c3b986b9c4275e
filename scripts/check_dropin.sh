#!/bin/bash
# Dev-container check (needs /root/reference; nothing from it is copied into this repo):
# an UNMODIFIED copy of the reference's Python package, with its pybind11 extension replaced by the one-line shim of
# INTEGRATION.md, imports and passes the reference's own CLI / config tests; on a GPU box the graph tests
# (tests/smoke/test_graph.py, test_outputs.py) run on top of libseqwin_hip.so the same way.
set -e
REF=${REF:-/root/reference}
REPO=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cp -r "$REF/src/seqwin" "$T/seqwin"
rm -f "$T"/seqwin/graph/_core*.so
printf 'from seqwin_amd._core import _build_native, _get_penalty_native, _filter_kmers_native  # noqa: F401\n' > "$T/seqwin/graph/_core.py"
cp -r "$REF/tests/smoke" "$T/smoke"
cd "$T"
export PYTHONPATH="$T:$REPO"
python -c "import seqwin; from seqwin.graph import KmerGraph, _get_penalty, _filter_kmers; import seqwin.graph._core as c; print('seqwin', seqwin.__version__, 'on', c._build_native.__module__)"
TESTS="smoke/test_config.py smoke/test_cli.py"
if python -c "from seqwin_amd._lib import lib; import sys; sys.exit(0 if lib.sw_device_count() > 0 else 1)"; then
    TESTS="smoke"     # with a GPU: the reference's whole smoke suite, incl. test_graph.py and the golden graph.npz / signatures
fi
python -m pytest -q $TESTS -p no:cacheprovider 2>&1 | tail -3
rm -rf "$T"
