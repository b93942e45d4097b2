"""JODO on QM9, unconditional (BASELINE configs 1 and 2)."""
from ._common import build


def get_config():
    return build()
