"""CPU-only: the spherical-shell generator (hyteg_amd/meshgen.py; MeshInfo::meshSphericalShell of the reference, BASELINE
config 5's mesh).  Counts as MeshGenSphericalShell.cpp (10 (ntan-1)^2 + 2 nodes per layer), volume -> shell volume, conformity:
every inner triangle belongs to exactly two tetrahedra, the boundary triangles form the two spheres."""
import itertools
import math
from collections import Counter
from pathlib import Path

import numpy as np
import pytest

from hyteg_amd import meshgen

MESHES = Path(__file__).resolve().parent.parent / "hyteg_amd" / "data" / "meshes"


def _volumes(v, c):
    a, b, cc, d = (v[c[:, k]] for k in range(4))
    return np.abs(np.einsum("ij,ij->i", np.cross(b - a, cc - a), d - a)) / 6.0


@pytest.mark.parametrize("ntan,layers", [(2, [1.0, 2.0, 3.0]), (3, [1.0, 1.5]), (5, [0.55, 0.7, 0.85, 1.0])])
def test_shell_mesh_is_conforming_and_fills_the_shell(ntan, layers):
    v, c = meshgen.spherical_shell(ntan, layers)
    n = ntan - 1
    assert len(v) == (10 * n * n + 2) * len(layers)  # MeshGenSphericalShell.cpp: nodes per layer
    assert len(c) == 20 * n * n * 3 * (len(layers) - 1)
    r = np.linalg.norm(v, axis=1)
    assert np.allclose(np.sort(np.unique(np.round(r, 12))), layers)
    vol = _volumes(v, c)
    assert vol.min() > 0.0
    exact = 4.0 / 3.0 * math.pi * (layers[-1] ** 3 - layers[0] ** 3)
    # a polyhedral shell: the volume converges to the ball shell's from below as ntan grows
    assert 0.55 * exact < vol.sum() < exact
    faces = Counter(tuple(sorted(t[list(f)])) for t in c for f in itertools.combinations(range(4), 3))
    assert set(faces.values()) <= {1, 2}
    boundary = [f for f, k in faces.items() if k == 1]
    assert len(boundary) == 2 * 20 * n * n  # inner and outer sphere
    rb = np.round(r[np.array(boundary)], 12)
    assert np.all((rb == layers[0]).all(axis=1) | (rb == layers[-1]).all(axis=1))
    # Euler characteristic of a thick shell (S^2 x [0,1]): V - E + F - C = 2
    edges = {tuple(sorted(t[list(e)])) for t in c for e in itertools.combinations(range(4), 2)}
    assert len(v) - len(edges) + len(faces) - len(c) == 2


def test_ntan2_node_set_is_the_icosahedron_of_the_reference():
    """MeshGenSphericalShell.cpp:805-836: poles on the z-axis, two rings of five at cos(colatitude) = 1 / sqrt(5)"""
    v, faces = meshgen.icosahedron()
    assert np.allclose(v[0], (0, 0, 1)) and np.allclose(v[11], (0, 0, -1))
    assert np.allclose(v[1:6, 2], 1.0 / math.sqrt(5.0)) and np.allclose(v[6:11, 2], -1.0 / math.sqrt(5.0))
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0)
    assert np.allclose(v[6], (2.0 / math.sqrt(5.0), 0.0, -1.0 / math.sqrt(5.0)))


def test_committed_mesh_file_is_what_the_generator_writes(tmp_path):
    v, c = meshgen.spherical_shell(2, [1.0, 2.0, 3.0])
    out = tmp_path / "shell.msh"
    meshgen.write_msh(out, v, c)
    assert out.read_text() == (MESHES / "spherical_shell_ntan2_3layers.msh").read_text()
    assert len(v) == 36 and len(c) == 120
