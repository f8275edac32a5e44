"""MI355X-native drop-in for the greedy sparse-NNLS path of ``bayesiancoresets``:

    import bayesiancoresets_amd as bc
    alg = bc.HilbertCoreset(X, projector, snnls=bc.snnls.GIGA)
    alg.build(1000); wts, pts, idcs = alg.get()

Namespace mirrors bayesiancoresets/__init__.py:1-2 for the components on that path
(SURVEY.md section 8); the batch pseudocoreset (BPSVI) is out of scope of this engine."""
from .coreset import Coreset, HilbertCoreset, UniformSamplingCoreset, SparseVICoreset, ShardedHilbertCoreset
from .projector import BlackBoxProjector, Projector, DeviceProjector
from . import snnls
from . import util

__version__ = "0.1.0"
