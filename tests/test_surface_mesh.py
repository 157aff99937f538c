"""CPU: the build's own iso-surface mesher (spurfies_amd/utils/surface.py:triangulate, marching tetrahedra) on analytic signed
distance volumes: closed 2-manifold, on the surface, outward normals, area; undefined (no-neighbour) regions are left open."""
import numpy as np

from spurfies_amd.utils import surface


def _sphere_volume(n=40, r=0.5, b=0.7):
    grid = surface.get_grid(None, n, input_min=np.array([-b, -b, -b]), input_max=np.array([b, b, b]), eps=0.0)
    p = grid["grid_points"].numpy().astype(np.float64)
    x, y, z = grid["xyz"]
    vol = (np.linalg.norm(p, axis=1) - r).astype(np.float32).reshape(len(y), len(x), len(z))
    return vol, grid


def _edges(faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    return np.sort(e, 1)


def test_sphere_is_a_closed_outward_oriented_manifold():
    vol, grid = _sphere_volume()
    verts, faces = surface.triangulate(vol, grid)
    assert len(verts) > 1000 and len(faces) > 2000
    h = grid["xyz"][0][1] - grid["xyz"][0][0]
    assert np.abs(np.linalg.norm(verts, axis=1) - 0.5).max() < 0.6 * h * h / 0.5        # linear interpolation of a curved field
    uniq, counts = np.unique(_edges(faces), axis=0, return_counts=True)
    assert (counts == 2).all()                                                         # watertight: every edge in exactly two faces
    assert len(verts) - len(uniq) + len(faces) == 2                                    # Euler characteristic of a sphere
    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    nrm = np.cross(p1 - p0, p2 - p0)
    assert ((nrm * (p0 + p1 + p2)).sum(1) > 0).all()                                   # normals point away from the centre
    area = 0.5 * np.linalg.norm(nrm, axis=1).sum()
    assert abs(area - 4 * np.pi * 0.25) < 0.01 * 4 * np.pi * 0.25
    # the welded vertices on cube edges are the points surface_points() reports
    sp = surface.surface_points(vol, grid)
    assert surface.chamfer(sp, verts)[1] < 1e-9                                        # every edge crossing is a mesh vertex


def test_cells_touching_undefined_samples_are_skipped(tmp_path):
    vol, grid = _sphere_volume()
    x, y, z = grid["xyz"]
    gx = np.meshgrid(x, y, z)[0]
    vol = vol.copy()
    vol[gx > 0.2] = surface.SDF_FILL                                                   # no neighbour there (get_sdf_eval's 1000)
    verts, faces = surface.triangulate(vol, grid)
    h = x[1] - x[0]
    assert len(faces) > 500 and verts[:, 0].max() <= 0.2 + 1e-9
    uniq, counts = np.unique(_edges(faces), axis=0, return_counts=True)
    assert set(np.unique(counts)) <= {1, 2} and (counts == 1).sum() > 10               # an open rim, nothing non-manifold
    path = tmp_path / "m.ply"
    surface.write_ply(path, verts, faces)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert f"element vertex {len(verts)}".encode() in head and f"element face {len(faces)}".encode() in head
    assert len(body) == 12 * len(verts) + 13 * len(faces)
    got = np.frombuffer(body[:12 * len(verts)], dtype="<f4").reshape(-1, 3)
    np.testing.assert_allclose(got, verts.astype(np.float32))
