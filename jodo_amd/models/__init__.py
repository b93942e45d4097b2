"""Model registry of the hot path.  Importing this package registers the HIP-backed
'DGT_concat' and 'cond_DGT_concat' (reference: models/__init__.py:1-4)."""
from .utils import create_model, register_model, get_model_class  # noqa: F401
from .node_distribution import get_node_dist, DistributionNodes, load_dataset_info  # noqa: F401
from .dgt import DGT_concat, Cond_DGT_concat  # noqa: F401
from .init_utils import deterministic_init_  # noqa: F401
