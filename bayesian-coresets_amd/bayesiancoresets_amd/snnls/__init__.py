from .snnls import SparseNNLS
from .giga import GIGA
from .frankwolfe import FrankWolfe
from .orthopursuit import OrthoPursuit
from .sampling import ImportanceSampling, UniformSampling
