"""Coreset classes of the drop-in surface (reference package ``bayesiancoresets.coreset``), plus the
row-sharded Hilbert coreset of this engine."""
from . import coreset as _base
from . import hilbert as _hilbert
from . import sampling as _sampling
from . import sparsevi as _sparsevi
from . import sharded_hilbert as _sharded

Coreset = _base.Coreset
HilbertCoreset = _hilbert.HilbertCoreset
UniformSamplingCoreset = _sampling.UniformSamplingCoreset
SparseVICoreset = _sparsevi.SparseVICoreset
ShardedHilbertCoreset = _sharded.ShardedHilbertCoreset



class BatchPSVICoreset(Coreset):
    """The name the reference exports (bayesiancoresets/__init__.py:1, coreset/bpsvi.py:6-64).  The batch pseudocoreset
    optimises weights AND pseudo-points by ADAM with no greedy scan: it is not on the path this engine accelerates
    (DESIGN.md section 7), so constructing it says so instead of an AttributeError."""

    def __init__(self, *args, **kw):
        raise NotImplementedError("bayesiancoresets_amd does not provide BatchPSVICoreset (out of the accelerated greedy / "
                                  "SparseVI path; the reference package's bayesiancoresets.BatchPSVICoreset serves it on the host)")


__all__ = ["Coreset", "HilbertCoreset", "UniformSamplingCoreset", "SparseVICoreset", "BatchPSVICoreset", "ShardedHilbertCoreset"]
