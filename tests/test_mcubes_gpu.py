"""GPU marching cubes (csrc/mcubes.hip) against the CPU oracle and, at the benchmark's 128^3, through order-independent
properties (closed oriented surface, Euler characteristic, enclosed volume)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from test_mcubes_cpu import _sphere, _torus   # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_gpu_mesh_equals_oracle_on_small_grids(dev):
    from oracle import mc_oracle as MO
    from shapeformer_amd import mcubes
    rng = np.random.RandomState(1)
    noise = np.zeros((13, 13, 13), np.float32)
    noise[1:-1, 1:-1, 1:-1] = rng.rand(11, 11, 11)
    open_surface = rng.rand(10, 10, 10).astype(np.float32)        # surface runs into the border: open mesh, same answer
    for occ, bbox in ((_sphere(16).astype(np.float32), ((-1, -1, -1), (1, 1, 1))),
                      (_torus(20).astype(np.float32), ((-1, -1, -1), (1, 1, 1))),
                      (noise, ((-2, -1, 0), (2, 1, 3))), (open_surface, ((-1, -1, -1), (1, 1, 1)))):
        v, f, voff, toff = mcubes.marching_cubes_dev(torch.from_numpy(occ)[None].to(dev), 0.5, bbox)
        vo, fo = MO.marching_cubes(occ, 0.5, bbox)
        assert voff.tolist() == [0, len(vo)] and toff.tolist() == [0, len(fo)]
        assert np.array_equal(f.cpu().numpy(), fo)                          # indices: bit-exact
        assert np.abs(v.cpu().numpy() - vo).max() < 1e-6                    # positions: f32, fused multiply-add on the GPU


def test_batched_offsets_and_local_indices(dev):
    from oracle import mc_oracle as MO
    from shapeformer_amd import mcubes
    grids = [_sphere(14, 0.5).astype(np.float32), np.zeros((14, 14, 14), np.float32), _torus(14, 0.5, 0.25).astype(np.float32)]
    v, f, voff, toff = mcubes.marching_cubes_dev(torch.from_numpy(np.stack(grids)).to(dev), 0.5)
    assert voff[1] == voff[2] and toff[1] == toff[2]                      # the empty grid contributes nothing
    for b, g in enumerate(grids):
        vo, fo = MO.marching_cubes(g, 0.5)
        assert np.array_equal(f[toff[b]:toff[b + 1]].cpu().numpy(), fo)
        assert np.abs(v[voff[b]:voff[b + 1]].cpu().numpy() - vo).max() < 1e-6 if len(vo) else True


def test_benchmark_resolution_properties(dev):
    from oracle import mc_oracle as MO
    from shapeformer_amd import mcubes
    occ = np.stack([_sphere(128, 0.6), _torus(128)]).astype(np.float32)
    v, f, voff, toff = mcubes.marching_cubes_dev(torch.from_numpy(occ).to(dev), 0.5)
    v, f = v.cpu().numpy(), f.cpu().numpy()
    vs, fs = v[voff[0]:voff[1]], f[toff[0]:toff[1]]
    assert MO.edge_use(fs) and MO.euler_characteristic(vs, fs) == 2
    assert abs(MO.signed_volume(vs, fs) / (4 / 3 * np.pi * 0.6 ** 3) - 1) < 2e-3
    vt, ft = v[voff[1]:voff[2]], f[toff[1]:toff[2]]
    assert MO.edge_use(ft) and MO.euler_characteristic(vt, ft) == 0
    assert abs(MO.signed_volume(vt, ft) / (2 * np.pi ** 2 * 0.55 * 0.22 ** 2) - 1) < 5e-3
    # array2mesh surface: bbox from coords, float64 / int outputs as the reference returns
    x = np.linspace(-1, 1, 128)
    coords = np.stack(np.meshgrid(x, x, x, indexing="ij"), -1).reshape(-1, 3)
    va, fa = mcubes.array2mesh(occ[0].reshape(-1), thresh=0.5, coords=coords)
    assert va.dtype == np.float64 and np.array_equal(fa, fs) and np.allclose(va, vs)


def test_mesh_of_a_reconstructed_shape(dev):
    """End of the path: the occupancy grid decode_index leaves in HBM -> mesh, equal to the oracle on the same grid."""
    from oracle import mc_oracle as MO
    from shapeformer_amd import mcubes, synthetic, weights as W
    from shapeformer_amd.vqdif import VQDIF
    vq = VQDIF(W.make_state_dict(W.vqdif_spec(16)), res=16, device=dev)
    cloud = torch.from_numpy(synthetic.make_batch(5, 1, n_full=8192, n_partial=4096)["Xbd"]).to(dev)
    q = vq.quantize_cloud_dev(cloud)[0]
    occ = vq.decode_index(q, grid_Q=32, sigmoid=True)["logits"].reshape(1, 32, 32, 32)
    iso = float(occ.median())                   # hash weights: a level the field is sure to cross
    v, f, voff, toff = mcubes.marching_cubes_dev(occ, iso)
    vo, fo = MO.marching_cubes(occ[0].cpu().numpy(), iso)
    assert len(fo) > 100 and np.array_equal(f.cpu().numpy(), fo) and np.abs(v.cpu().numpy() - vo).max() < 1e-6


def test_config4_resolution_256(dev):
    """BASELINE config 4 queries a 256^3 lattice: 50 M grid edges in one call (index scratch 268 MB), same properties."""
    from oracle import mc_oracle as MO
    from shapeformer_amd import mcubes
    occ = torch.from_numpy(_sphere(256, 0.7).astype(np.float32))[None].to(dev)
    v, f, voff, toff = mcubes.marching_cubes_dev(occ, 0.5)
    v, f = v.cpu().numpy(), f.cpu().numpy()
    assert MO.edge_use(f) and MO.euler_characteristic(v, f) == 2
    assert abs(MO.signed_volume(v, f) / (4 / 3 * np.pi * 0.7 ** 3) - 1) < 1e-3
    r = np.linalg.norm(v, axis=1)
    assert abs(r.mean() - 0.7) < 1e-3 and r.std() < 2e-3
