"""Sparse non-negative least-squares solvers (reference package ``bayesiancoresets.snnls``): the three greedy
device solvers and the two host-side sampling baselines."""
from . import snnls as _core
from . import giga as _giga
from . import frankwolfe as _fw
from . import orthopursuit as _omp
from . import sampling as _sampling

SparseNNLS = _core.SparseNNLS
DeviceSparseNNLS = _core.DeviceSparseNNLS
GIGA = _giga.GIGA
FrankWolfe = _fw.FrankWolfe
OrthoPursuit = _omp.OrthoPursuit
ImportanceSampling = _sampling.ImportanceSampling
UniformSampling = _sampling.UniformSampling

__all__ = ["SparseNNLS", "GIGA", "FrankWolfe", "OrthoPursuit", "ImportanceSampling", "UniformSampling"]
