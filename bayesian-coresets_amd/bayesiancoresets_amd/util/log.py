"""Logging conventions of the reference (bayesiancoresets/util/log.py:5-7,13-42):
root logger, one stderr handler whose format names the emitting object via an
``id`` field, default level ERROR, ``set_verbosity(str)``."""
import logging
import secrets
import sys

_LEVELS = {
    "error": logging.ERROR,
    "warning": logging.WARNING,
    "critical": logging.CRITICAL,
    "info": logging.INFO,
    "debug": logging.DEBUG,
    "notset": logging.NOTSET,
}
FORMAT = "%(levelname)s - %(id)s.%(funcName)s(): %(message)s"


def set_verbosity(verb):
    logging.getLogger().setLevel(_LEVELS[verb])


class _IdFilter(logging.Filter):
    """Records from other libraries have no ``id``; give them one so the shared
    root handler never fails to format."""

    def filter(self, record):
        if not hasattr(record, "id"):
            record.id = record.name
        return True


def _install_handler():
    root = logging.getLogger()
    for h in root.handlers:
        if getattr(h, "_bcx_handler", False):
            return
    h = logging.StreamHandler(sys.stderr)
    h.setFormatter(logging.Formatter(FORMAT))
    h.addFilter(_IdFilter())
    h._bcx_handler = True
    root.addHandler(h)
    root.setLevel(_LEVELS["error"])


def object_logger(obj):
    """(alg_name, LoggerAdapter) pair: ``ClassName-<3 random hex bytes>`` as in
    snnls.py:10-11 / coreset.py:9-10."""
    name = obj.__class__.__name__ + "-" + secrets.token_hex(3)
    return name, logging.LoggerAdapter(logging.getLogger(), {"id": name})


_install_handler()
