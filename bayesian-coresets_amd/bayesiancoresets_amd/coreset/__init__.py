from .coreset import Coreset
from .hilbert import HilbertCoreset
from .sampling import UniformSamplingCoreset
from .sparsevi import SparseVICoreset
from .sharded_hilbert import ShardedHilbertCoreset
