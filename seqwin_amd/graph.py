"""Mirror of the reference's ``seqwin.graph`` operator interface (src/seqwin/graph/__init__.py).

Same names, argument meaning and error behaviour: ``KmerGraph`` (:61-146), ``_get_penalty`` (:149-171),
``_filter_kmers`` (:174-196) and the three structured dtypes (:38-58) -- over the MI355X
implementation in :mod:`seqwin_amd._core`.  See INTEGRATION.md for pointing an unmodified Seqwin
checkout at it.
"""
from __future__ import annotations

from collections.abc import Iterable
from pathlib import Path

import numpy as np
from numpy.typing import NDArray

from ._core import (EDGE_DTYPE, KMER_DTYPE, NODE_DTYPE, _build_native, _filter_kmers_native,
                    _get_penalty_native)

__all__ = ["KMER_DTYPE", "NODE_DTYPE", "EDGE_DTYPE", "KmerGraph", "_get_penalty", "_filter_kmers"]


class KmerGraph:
    """The minimizer graph (reference src/seqwin/graph/__init__.py:61-146).

    Attributes: ``kmers`` (KMER_DTYPE, occurrences grouped and sorted by hash, ``record_idx``
    nondecreasing inside a node), ``nodes`` (NODE_DTYPE, sorted by hash, ``[start, stop)`` into
    ``kmers``), ``edges`` (EDGE_DTYPE, sorted by (first, second), weight = number of assemblies where
    the two minimizers are adjacent), ``record_offsets`` (uint32, cumulative records per assembly) and
    ``record_ids`` (one tuple of FASTA ids per assembly).
    """
    __slots__ = ("kmers", "nodes", "edges", "record_offsets", "record_ids")
    kmers: NDArray[np.void]
    nodes: NDArray[np.void]
    edges: NDArray[np.void]
    record_offsets: NDArray[np.uint32]
    record_ids: list[tuple[str, ...]]

    def __init__(self, assembly_paths: Iterable[Path], kmerlen: int, windowsize: int,
                 low_memory: bool = False, n_cpu: int = 1) -> None:
        self.kmers, self.nodes, self.edges, self.record_offsets, self.record_ids = _build_native(
            list(str(p) for p in assembly_paths), int(kmerlen), int(windowsize), int(n_cpu), bool(low_memory))


def _get_penalty(kmers: NDArray[np.void], nodes: NDArray[np.void], record_offsets: NDArray[np.uint32],
                 is_targets: Iterable[bool], n_cpu: int = 1) -> None:
    """Populate ``nodes['n_tar','n_neg','penalty']`` in place (reference :149-171)."""
    _get_penalty_native(kmers, nodes, record_offsets, np.asarray(is_targets, dtype=np.bool_, order="C"), int(n_cpu))


def _filter_kmers(kmers: NDArray[np.void], nodes: NDArray[np.void], used_hashes) -> tuple[NDArray[np.void], NDArray[np.void]]:
    """Keep the nodes whose hash is in ``used_hashes`` and their k-mers, re-basing ranges (reference :174-196)."""
    return _filter_kmers_native(kmers, nodes, used_hashes)
