"""seqwin_amd -- MI355X (gfx950) implementation of Seqwin's minimizer-index hot path.

Only what that path needs lives here:

* ``csrc/``      hand-written HIP kernels + the C ABI (``include/seqwin_hip.h`` -> ``libseqwin_hip.so``)
* ``_core``      ctypes drop-in for the reference's pybind11 module ``seqwin.graph._core``
* ``graph``      mirror of ``seqwin.graph`` (KmerGraph, _get_penalty, _filter_kmers, dtypes)
* ``device``     device-resident batches / indexes (bench, multi-GPU)
* ``dist``       one-process-per-GPU sharding and the RCCL merge

Importing the package loads the shared library and fails loudly when it is missing.
"""
from .graph import EDGE_DTYPE, KMER_DTYPE, NODE_DTYPE, KmerGraph, _filter_kmers, _get_penalty  # noqa: F401

__version__ = "0.1.0"
