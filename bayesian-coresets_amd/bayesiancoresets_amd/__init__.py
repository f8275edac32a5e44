"""MI355X-native drop-in for the greedy sparse-NNLS path of ``bayesiancoresets``:

    import bayesiancoresets_amd as bc
    alg = bc.HilbertCoreset(X, projector, snnls=bc.snnls.GIGA)
    alg.build(1000); wts, pts, idcs = alg.get()

Namespace mirrors bayesiancoresets/__init__.py:1-2 for the classes on the greedy / SparseVI path;
``BatchPSVICoreset`` is out of scope: the name exists and raises NotImplementedError."""
from .coreset import Coreset, HilbertCoreset, UniformSamplingCoreset, SparseVICoreset, BatchPSVICoreset, ShardedHilbertCoreset
from .projector import BlackBoxProjector, Projector, DeviceProjector
from .linreg_sampler import LinregPosteriorSampler
from .laplace_sampler import LaplacePosteriorSampler
from . import snnls
from . import util

__version__ = "0.1.0"
