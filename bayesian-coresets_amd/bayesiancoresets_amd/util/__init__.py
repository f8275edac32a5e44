"""bc.util namespace: tolerance used by the numeric-limit checks, logging verbosity, projected ADAM.
(reference surface: bayesiancoresets/util/__init__.py:1-8)"""
from . import errors
from .log import set_verbosity
from .opt import nn_opt

_DEFAULT_TOL = 1e-12
TOL = _DEFAULT_TOL   # read by the solvers at build() time as ``util.TOL``


def set_tolerance(tol):
    """Rebind the module-level ``TOL`` (solvers pick it up on their next build()/optimize())."""
    globals()["TOL"] = tol
