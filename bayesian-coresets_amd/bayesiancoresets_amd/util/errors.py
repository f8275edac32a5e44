"""Error type of the library (reference: bayesiancoresets/util/errors.py:1)."""


class NumericalPrecisionError(Exception):
    """Raised / caught where the reference signals that numeric precision ran out."""
