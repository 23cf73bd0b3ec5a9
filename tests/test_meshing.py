"""Host-side iso-surface extractor (endosurf_amd/meshing.py): level set of an analytic sphere field.  CPU only."""
import numpy as np

from endosurf_amd.meshing import iso_surface, marching_tetrahedra


def _sphere_field(R=28, r=0.6):
    ax = np.linspace(-1, 1, R)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    return np.sqrt(x * x + y * y + z * z) - r, ax


def test_sphere_is_closed_and_on_the_level_set():
    u, ax = _sphere_field()
    v, f = marching_tetrahedra(u, 0.0)
    assert v.shape[1] == 3 and f.shape[1] == 3 and len(f) > 500
    p = v / (len(ax) - 1) * 2 - 1                                   # index -> world coordinates
    rad = np.linalg.norm(p, axis=1)
    assert np.abs(rad - 0.6).max() < 0.01                           # linear interpolation error of a curved field
    # watertight: every edge is shared by exactly two triangles
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()
    # outward orientation (normals along +grad u) and area ~ 4 pi r^2
    p0, p1, p2 = p[f[:, 0]], p[f[:, 1]], p[f[:, 2]]
    n = np.cross(p1 - p0, p2 - p0)
    assert (np.einsum("ij,ij->i", n, (p0 + p1 + p2) / 3) > 0).all()
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    assert abs(area - 4 * np.pi * 0.36) < 0.02 * 4 * np.pi * 0.36
    # Euler characteristic of a sphere
    assert len(v) - len(e) // 2 + len(f) == 2


def test_threshold_and_empty():
    u, _ = _sphere_field(16)
    v0, _ = marching_tetrahedra(u, 0.0)
    v1, _ = marching_tetrahedra(u, 0.2)
    c = (16 - 1) / 2
    assert np.linalg.norm(v1 - c, axis=1).mean() > np.linalg.norm(v0 - c, axis=1).mean()
    v, f = marching_tetrahedra(u, -5.0)
    assert len(v) == 0 and len(f) == 0
    v, f = iso_surface(u, 0.0)
    assert len(f) > 0
