"""Thick spherical shell meshes of tetrahedra: an own, minimal generator in the spirit of HyTeG's MeshInfo::meshSphericalShell
( ntan, layers ) (src/hyteg/mesh/MeshGenSphericalShell.cpp; used by apps/stokesSphere/StokesSphere.cpp:109 -- BASELINE config 5).

What is the same as the reference's: the icosahedron the shell is built on (north pole (0,0,1), south pole (0,0,-1), the two
rings of five vertices at colatitude w with cos w = 1/sqrt(5), the upper one rotated by pi/5: MeshGenSphericalShell.cpp:805-836),
`ntan` nodes along every edge of its 20 faces (10 (ntan-1)^2 + 2 nodes per spherical layer), one layer of nodes per entry of
`layers` (radii, ascending), prisms between consecutive layers cut into tetrahedra, affine cells (no blending map).
What differs: for ntan > 2 the tangential nodes are the normalised barycentric subdivision points of the icosahedron's faces
(the reference bisects great-circle arcs inside its ten diamonds), and every prism is cut into 3 tetrahedra along the diagonals
that start at the lower-numbered vertex (conforming by construction), where the reference cuts its hexahedral cells into 6.
For ntan = 2 (the app's default parameter file) the node set is identical to the reference's: the 12 icosahedron vertices per
layer."""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np


def icosahedron():
    """vertices (12 x 3, unit sphere) and faces (20 x 3) with the reference's vertex placement"""
    fifthpi = 0.4 * math.asin(1.0)
    w = 2.0 * math.acos(1.0 / (2.0 * math.sin(fifthpi)))
    v = np.zeros((12, 3))
    v[0] = (0.0, 0.0, 1.0)
    v[11] = (0.0, 0.0, -1.0)
    for k in range(1, 6):
        phi = 2.0 * (k - 0.5) * fifthpi
        v[k] = (math.sin(w) * math.cos(phi), math.sin(w) * math.sin(phi), math.cos(w))
        phi = 2.0 * (k - 1) * fifthpi
        v[k + 5] = (math.sin(w) * math.cos(phi), math.sin(w) * math.sin(phi), -math.cos(w))
    # faces = triples of mutually adjacent vertices (edge length of the unit icosahedron: 4 / sqrt(10 + 2 sqrt 5))
    edge = 4.0 / math.sqrt(10.0 + 2.0 * math.sqrt(5.0))
    adj = np.abs(np.linalg.norm(v[:, None, :] - v[None, :, :], axis=2) - edge) < 1e-9
    faces = [(a, b, c) for a in range(12) for b in range(a + 1, 12) for c in range(b + 1, 12) if adj[a, b] and adj[b, c] and adj[a, c]]
    assert len(faces) == 20 and adj.sum() == 60
    return v, faces


def spherical_surface(ntan: int):
    """nodes on the unit sphere (n x 3) and triangles (m x 3): every icosahedron face subdivided into (ntan - 1)^2 triangles"""
    if ntan < 2:
        raise ValueError("ntan >= 2 (nodes along an icosahedron edge)")
    v, faces = icosahedron()
    n = ntan - 1
    nodes, index = [tuple(p) for p in v], {}

    def node(a, b, c, i, j, k):
        # barycentric point (i, j, k) / n of face (a, b, c); shared between faces through its sorted (vertex, weight) pairs
        key = tuple(sorted((x, w_) for x, w_ in ((a, i), (b, j), (c, k)) if w_ > 0))
        if len(key) == 1:
            return key[0][0]
        if key not in index:
            p = (i * v[a] + j * v[b] + k * v[c]) / n
            index[key] = len(nodes)
            nodes.append(tuple(p / np.linalg.norm(p)))
        return index[key]

    tris = []
    for a, b, c in faces:
        for i in range(n):
            for j in range(n - i):
                k = n - i - j
                p0, p1, p2 = node(a, b, c, i, j, k), node(a, b, c, i + 1, j, k - 1), node(a, b, c, i, j + 1, k - 1)
                tris.append((p0, p1, p2))
                if i + j < n - 1:
                    tris.append((p1, node(a, b, c, i + 1, j + 1, k - 2), p2))
    nodes = np.array(nodes)
    assert len(nodes) == 10 * n * n + 2 and len(tris) == 20 * n * n
    return nodes, tris


def spherical_shell(ntan: int, layers):
    """vertices (n x 3) and tetrahedra (m x 4, 0-based) of the shell between the radii `layers` (ascending)"""
    layers = [float(r) for r in layers]
    if len(layers) < 2 or any(b <= a for a, b in zip(layers, layers[1:])) or layers[0] <= 0.0:
        raise ValueError("layers: at least two positive radii, ascending")
    surf, tris = spherical_surface(ntan)
    ns = len(surf)
    vertices = np.concatenate([r * surf for r in layers])
    cells = []
    for lay in range(len(layers) - 1):
        lo, hi = lay * ns, (lay + 1) * ns
        for t in tris:
            v0, v1, v2 = sorted(t)  # the quad faces' diagonals start at the lower-numbered vertex: conforming across prisms
            a, b, c, A, B, C = lo + v0, lo + v1, lo + v2, hi + v0, hi + v1, hi + v2
            cells += [(a, b, c, C), (a, b, C, B), (a, B, C, A)]
    return vertices, np.array(cells, dtype=np.int64)


def write_msh(path, vertices, cells) -> None:
    """Gmsh 2.2 ASCII (the format of the reference's data/meshes/3D/*.msh)"""
    lines = ["$MeshFormat", "2.2 0 8", "$EndMeshFormat", "$Nodes", str(len(vertices))]
    lines += [f"{i + 1} {repr(float(p[0]))} {repr(float(p[1]))} {repr(float(p[2]))}" for i, p in enumerate(vertices)]
    lines += ["$EndNodes", "$Elements", str(len(cells))]
    lines += [f"{i + 1} 4 2 0 0 {c[0] + 1} {c[1] + 1} {c[2] + 1} {c[3] + 1}" for i, c in enumerate(cells)]
    lines += ["$EndElements", ""]
    Path(path).write_text("\n".join(lines))


if __name__ == "__main__":
    import sys

    ntan = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    radii = [float(x) for x in sys.argv[2:]] or [1.0, 2.0, 3.0]
    out = Path(__file__).resolve().parent / "data" / "meshes" / f"spherical_shell_ntan{ntan}_{len(radii)}layers.msh"
    v, c = spherical_shell(ntan, radii)
    write_msh(out, v, c)
    print(f"{out}: {len(v)} vertices, {len(c)} tetrahedra")
