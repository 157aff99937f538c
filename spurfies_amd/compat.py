"""Register this package's mirrors under the reference's module names (`spurfies.*`), so that
`from spurfies.model.pointneus_disent import PointVolSDF` and the string lookups of
spurfies/train.py:118-120 (`utils.get_class(conf.get_string("train.model_class"))`) resolve here."""
from __future__ import annotations

import importlib
import sys
import types

_MAP = {
    "spurfies.model.pointneus_disent": "spurfies_amd.model.pointneus_disent",
    "spurfies.model.ray_sampler": "spurfies_amd.model.ray_sampler",
    "spurfies.model.density": "spurfies_amd.model.density",
    "spurfies.model.embedder": "spurfies_amd.model.embedder",
    "spurfies.model.utils": "spurfies_amd.model.utils",
    "spurfies.model.loss": "spurfies_amd.model.loss",
    "spurfies.utils.rend_util": "spurfies_amd.utils.rend_util",
    "spurfies.utils.general": "spurfies_amd.utils.general",
    "spurfies.train": "spurfies_amd.train",
    "spurfies.feat_utils": "spurfies_amd.feat_utils",
}


def register():
    for pkg in ("spurfies", "spurfies.model", "spurfies.utils"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = []  # mark as package
            sys.modules[pkg] = m
    for alias, target in _MAP.items():
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        parent, _, leaf = alias.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)
    sys.modules["spurfies"].model = sys.modules["spurfies.model"]
    sys.modules["spurfies"].utils = sys.modules["spurfies.utils"]
