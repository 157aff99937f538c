"""spurfies_amd — MI355X-native (gfx950, HIP) implementation of the per-ray
volumetric-rendering hot path of Spurfies, behind the reference's own Python API.

    import spurfies_amd
    spurfies_amd.install_dropin()          # `import torch_knnquery` / `import spurfies...` now resolve here
"""
from __future__ import annotations

import sys

__version__ = "0.1.0"


def install_dropin():
    """Register this package's mirrors under the reference's module names so that code written
    against the reference (`from torch_knnquery import VoxelGrid`,
    `spurfies.model.pointneus_disent.PointVolSDF`, ...) runs unchanged."""
    import importlib

    from . import torch_knnquery as tk

    sys.modules["torch_knnquery"] = tk
    try:
        compat = importlib.import_module("spurfies_amd.compat")
    except ImportError:
        return
    compat.register()
