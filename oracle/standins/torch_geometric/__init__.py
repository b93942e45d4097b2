"""Stand-in for torch_geometric (absent in this image). Test infrastructure only."""
from . import typing, utils, nn  # noqa: F401
