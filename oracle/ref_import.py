"""Import the upstream reference (pure Python, /root/reference) under oracle/standins.

TEST INFRASTRUCTURE ONLY — used by oracle/make_golden.py and `-m "not gpu"` calibration tests in the
build container.  /root/reference does not exist on the GPU box: everything that runs there uses the
committed fixtures (tests/golden/) and oracle/dgt_oracle.py instead.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("JODO_REFERENCE_ROOT", "/root/reference")
_STANDINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "standins")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "mol_gnn.py"))


class _Ref:
    """Lazy handle on the reference's modules (models, sampling, ...)."""

    def __init__(self):
        self._mods = {}

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        if name not in self._mods:
            self._mods[name] = _import(name)
        return self._mods[name]


def _import(name):
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    for p in (_STANDINS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    # our package must never shadow / be shadowed by the reference's top-level names
    return importlib.import_module(name)


def load_reference():
    """Returns a handle `ref` with ref.models, ref.sampling, ref.mix_dpm_solver, ref.diffusion,
    ref.utils, ref.configs.<name> importable."""
    return _Ref()


def reference_config(name):
    """name in {'vpsde_qm9_uncond_jodo','vpsde_geom_uncond_jodo','vpsde_qm9_cond_jodo'}"""
    if not reference_available():
        raise RuntimeError("reference not present")
    for p in (_STANDINS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    mod = importlib.import_module("configs." + name)
    return mod.get_config()
