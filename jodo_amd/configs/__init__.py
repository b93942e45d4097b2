"""Config surface of the three JODO experiments on the hot path (same keys and defaults as the
reference's configs/vpsde_qm9_uncond_jodo.py, vpsde_geom_uncond_jodo.py, vpsde_qm9_cond_jodo.py)."""
from . import vpsde_qm9_uncond_jodo, vpsde_geom_uncond_jodo, vpsde_qm9_cond_jodo  # noqa: F401


def get(name):
    import importlib
    return importlib.import_module(__name__ + '.' + name).get_config()
