"""pytest wiring: the ``gpu`` marker, import paths, shared fixtures.

``-m "not gpu"`` runs here (no GPU): oracle KATs, golden fixtures, C-ABI load/symbol checks, host logic,
gloo multi-process tests.  ``-m gpu`` runs on the MI355X box: HIP-vs-oracle parity through the C ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instant-nsr-pl_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (runs on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


NERF_GRID = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                 per_level_scale=1.447269237440378)
NEUS_GRID = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=32,
                 per_level_scale=1.3195079107728942)
DENSITY_MLP = dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64, n_hidden_layers=1)
COLOR_MLP = dict(otype="FullyFusedMLP", activation="ReLU", output_activation="Sigmoid", n_neurons=64, n_hidden_layers=2)
