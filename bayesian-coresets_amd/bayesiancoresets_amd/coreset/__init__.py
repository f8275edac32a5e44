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

__all__ = ["Coreset", "HilbertCoreset", "UniformSamplingCoreset", "SparseVICoreset", "ShardedHilbertCoreset"]
