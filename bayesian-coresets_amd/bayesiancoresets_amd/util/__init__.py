"""bc.util namespace (reference: bayesiancoresets/util/__init__.py:1-8)."""
from .opt import nn_opt
from .log import set_verbosity
from . import errors

TOL = 1e-12


def set_tolerance(tol):
    global TOL
    TOL = tol
