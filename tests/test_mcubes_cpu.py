"""Iso-surface step, CPU side: the marching-cubes oracle's order-independent properties, the
generated case table against the oracle's table-free tracing, and the mesh result formats (binary PLY, surface samples)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _grid(Q):
    x = np.linspace(-1, 1, Q)
    return np.meshgrid(x, x, x, indexing="ij")


def _sphere(Q, r=0.6, c=(0, 0, 0)):
    X, Y, Z = _grid(Q)
    d = np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2)
    return 1 / (1 + np.exp(10 * (d - r)))


def _torus(Q, R=0.55, r=0.22):
    X, Y, Z = _grid(Q)
    d = np.sqrt((np.sqrt(X ** 2 + Y ** 2) - R) ** 2 + Z ** 2)
    return 1 / (1 + np.exp(12 * (d - r)))


def test_generated_case_table_equals_table_free_tracing():
    from oracle import mc_oracle as MO
    from shapeformer_amd import mc_tables as MT
    ntri, tri = MT.tables()
    assert int(ntri.max()) <= MT.MAXT and int(ntri.sum()) == 820     # classic marching cubes: 820 triangles over 256 cases
    for ci in range(256):
        want = MO.pattern_triangles(ci)
        got = [tuple(int(x) for x in tri[ci, 3 * k:3 * k + 3]) for k in range(ntri[ci])]
        assert got == want, ci
        assert (tri[ci, 3 * ntri[ci]:] == 255).all()
    # the committed C table is the generator's output
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        MT.emit_inc(os.path.join(d, "t.h"))
        assert open(os.path.join(d, "t.h")).read() == open(os.path.join(ROOT, "shapeformer_amd", "csrc", "mc_table.h")).read()


def test_oracle_meshes_are_closed_oriented_and_have_the_right_topology():
    from oracle import mc_oracle as MO
    v, f = MO.marching_cubes(_sphere(18), 0.5)
    assert MO.edge_use(f) and MO.euler_characteristic(v, f) == 2
    vol = MO.signed_volume(v, f)
    assert abs(vol - 4 / 3 * np.pi * 0.6 ** 3) / (4 / 3 * np.pi * 0.6 ** 3) < 0.04 and vol > 0    # outward normals
    v, f = MO.marching_cubes(_torus(22), 0.5)
    assert MO.edge_use(f) and MO.euler_characteristic(v, f) == 0
    two = np.maximum(_sphere(20, 0.3, (-0.45, 0, 0)), _sphere(20, 0.3, (0.45, 0.1, 0)))
    v, f = MO.marching_cubes(two, 0.5)
    assert MO.edge_use(f) and MO.euler_characteristic(v, f) == 4
    # vertices sit on grid edges at the interpolated iso crossing, inside the bbox mapping of array2mesh
    v, f = MO.marching_cubes(_sphere(12), 0.5, bbox=((-2, -1, 0), (2, 1, 3)))
    assert v[:, 0].min() > -2 and v[:, 0].max() < 2 and v[:, 2].min() > 0 and v[:, 2].max() < 3
    g = (v - np.array([-2, -1, 0])) / np.array([4, 2, 3]) * 11
    on_edge = (np.abs(g - np.round(g)) < 1e-4).sum(1)
    assert (on_edge >= 2).all()


def test_oracle_random_field_with_ambiguous_cells_is_still_watertight():
    from oracle import mc_oracle as MO
    rng = np.random.RandomState(3)
    occ = np.zeros((14, 14, 14), np.float32)
    occ[1:-1, 1:-1, 1:-1] = rng.rand(12, 12, 12)      # white noise: every ambiguous pattern occurs; border is outside
    v, f = MO.marching_cubes(occ, 0.5)
    assert len(f) > 1000 and MO.edge_use(f)


def test_ply_round_trip_and_surface_samples(tmp_path):
    from oracle import mc_oracle as MO
    from shapeformer_amd import meshio
    v, f = MO.marching_cubes(_sphere(16), 0.5)
    p = meshio.write_mesh(str(tmp_path), v, f, "s0_mesh")
    assert p.endswith("meshes/s0_mesh.ply")
    head = open(p, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and b"property double x" in head
    v2, f2 = meshio.read_ply(p)
    assert np.array_equal(f2, f) and np.allclose(v2, v)
    pts = meshio.sample_mesh(v, f, 20000, rng=np.random.RandomState(0))
    r = np.linalg.norm(pts, axis=1)
    assert pts.shape == (20000, 3) and abs(r.mean() - 0.6) < 0.02 and r.std() < 0.02
    assert abs(pts.mean(0)).max() < 0.02                               # area-uniform: centroid at the origin
    # degenerate mesh -> the reference's dummy triangle (geoutil.write_mesh)
    p = meshio.write_mesh(str(tmp_path), v[:3], f[:1], "tiny")
    v3, f3 = meshio.read_ply(p)
    assert v3.shape == (1, 3) and f3.tolist() == [[0, 0, 0]]
