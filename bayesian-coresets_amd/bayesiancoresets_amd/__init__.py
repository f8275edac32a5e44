"""MI355X-native drop-in for the greedy sparse-NNLS path of ``bayesiancoresets``:

    import bayesiancoresets_amd as bc
    alg = bc.HilbertCoreset(X, projector, snnls=bc.snnls.GIGA)
    alg.build(1000); wts, pts, idcs = alg.get()

Namespace mirrors bayesiancoresets/__init__.py:1-2 (the batch pseudocoreset keeps the reference's host loop; its
N-sized column sums run on the device behind a DeviceProjector)."""
from .coreset import Coreset, HilbertCoreset, UniformSamplingCoreset, SparseVICoreset, BatchPSVICoreset, ShardedHilbertCoreset
from .projector import BlackBoxProjector, Projector, DeviceProjector
from . import snnls
from . import util

__version__ = "0.1.0"
