import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently skip: only add the skip when
    # the user did not explicitly select gpu tests.
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# the unit reference tetrahedron used throughout (data/meshes/3D/tet_1el.msh vertices)
REF_TET = ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))
# an irregular, non-axis-aligned tetrahedron (first cell of 3D/regular_octahedron_8el.msh: nodes 1 4 2 6)
OCT_TET = ((0.0, 0.0, 0.0), (0.5, 0.5, 1.0), (1.0, 0.0, 0.0), (0.5, 0.5, 0.0))
SKEW_TET = ((0.1, -0.2, 0.05), (1.3, 0.1, -0.1), (0.2, 0.9, 0.3), (-0.15, 0.25, 1.1))
