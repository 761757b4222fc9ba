"""GPU parity: fused SDF-query kernel (C ABI sfmi_sdf_query_f32) vs the CPU oracle (dec.py:62-100)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# fp32 tolerance: the kernel is an exact-f32 fma chain in a different summation order than MKL sgemm.
ATOL, RTOL = 2e-4, 1e-4


def _rand_grid(B, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 32, 64, 64, 64, generator=g)


@pytest.mark.parametrize("B,N", [(1, 1), (2, 31), (1, 32), (2, 1000), (3, 4097)])
def test_sdf_query_points(dev, vq16_sd, vq16_sd_t, B, N):
    from oracle import vqdif_oracle as O
    from shapeformer_amd import ops
    grid = _rand_grid(B, 1)
    g = torch.Generator().manual_seed(7 + N)
    xyz = (torch.rand(B, N, 3, generator=g) * 2.6 - 1.3)  # includes out-of-range points (border clamp)
    xyz[:, 0] = torch.tensor([1.0, -1.0, 0.0])[: 3]
    ref = O.sdf_query(vq16_sd_t, grid, xyz)
    wp = torch.from_numpy(ops.sdf_pack_weights(vq16_sd)).to(dev)
    grid_cl = grid.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    out = ops.sdf_query(xyz.to(dev), grid_cl, wp).cpu()
    assert out.shape == (B, N, 1)
    torch.testing.assert_close(out, ref, atol=ATOL, rtol=RTOL)


def test_sdf_query_grid_mode_matches_point_mode_and_oracle(dev, vq16_sd, vq16_sd_t):
    from oracle import vqdif_oracle as O
    from shapeformer_amd import ops
    Q, B = 40, 2
    grid = _rand_grid(B, 2)
    pts = torch.from_numpy(O.make_grid(Q))[None].expand(B, -1, -1).contiguous()
    axis = torch.from_numpy(np.linspace(-1.0, 1.0, Q).astype(np.float32))
    wp = torch.from_numpy(ops.sdf_pack_weights(vq16_sd)).to(dev)
    grid_cl = grid.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    a = ops.sdf_query_grid(axis.to(dev), grid_cl, wp).cpu()
    b = ops.sdf_query(pts.to(dev), grid_cl, wp).cpu()
    assert torch.equal(a, b)  # same arithmetic, only the coordinate source differs
    ref = O.sdf_query(vq16_sd_t, grid, pts)
    torch.testing.assert_close(a, ref, atol=ATOL, rtol=RTOL)
    s = ops.sdf_query_grid(axis.to(dev), grid_cl, wp, sigmoid=True).cpu()
    torch.testing.assert_close(s, torch.sigmoid(ref), atol=1e-5, rtol=1e-4)


def test_sdf_query_with_the_last_groupnorm_applied_in_the_kernel(dev, vq16_sd, vq16_sd_t):
    """sfmi_sdf_query_grid_aff_f32 (csrc/sdf_query.hip AFF): the decoder grid BEFORE its last GroupNorm + that GroupNorm's (B,32) affine ==
    the query on the affined grid (the trilinear 'border' weights sum to one) to fp32 rounding, == the oracle on the affined grid; a slab
    of planes of the affine form is bit-equal to the same planes of its whole-lattice call."""
    from oracle import vqdif_oracle as O
    from shapeformer_amd import ops
    Q, B = 24, 3
    grid = _rand_grid(B, 5)
    g = torch.Generator().manual_seed(7)
    sc, sh = torch.rand(B, 32, generator=g) + 0.5, torch.randn(B, 32, generator=g) * 0.3
    axis = torch.from_numpy(np.linspace(-1.0, 1.0, Q).astype(np.float32)).to(dev)
    wp = torch.from_numpy(ops.sdf_pack_weights(vq16_sd)).to(dev)
    raw_cl = grid.permute(0, 2, 3, 4, 1).contiguous().to(dev)
    aff_grid = grid * sc[:, :, None, None, None] + sh[:, :, None, None, None]
    a = ops.sdf_query_grid(axis, raw_cl, wp, affine=(sc.to(dev), sh.to(dev))).cpu()
    b = ops.sdf_query_grid(axis, aff_grid.permute(0, 2, 3, 4, 1).contiguous().to(dev), wp).cpu()
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    pts = torch.from_numpy(O.make_grid(Q))[None].expand(B, -1, -1).contiguous()
    torch.testing.assert_close(a, O.sdf_query(vq16_sd_t, aff_grid, pts), atol=ATOL, rtol=RTOL)
    part = ops.sdf_query_grid(axis, raw_cl, wp, affine=(sc.to(dev), sh.to(dev)), x_range=(5, 17)).cpu()
    assert torch.equal(part, a[:, 5 * Q * Q:17 * Q * Q])


def test_sdf_query_linearity_in_fc_out(dev, vq16_sd):
    """Size-independent property at the full 128^3 size: scaling fc_out scales (logit - bias)."""
    from shapeformer_amd import ops
    B, Q = 1, 128
    grid_cl = _rand_grid(B, 3).permute(0, 2, 3, 4, 1).contiguous().to(dev)
    axis = torch.linspace(-1, 1, Q).to(dev)
    sd2 = dict(vq16_sd)
    sd2["decoder.fc_out.weight"] = vq16_sd["decoder.fc_out.weight"] * 2.0
    sd2["decoder.fc_out.bias"] = vq16_sd["decoder.fc_out.bias"] * 2.0
    w1 = torch.from_numpy(ops.sdf_pack_weights(vq16_sd)).to(dev)
    w2 = torch.from_numpy(ops.sdf_pack_weights(sd2)).to(dev)
    a = ops.sdf_query_grid(axis, grid_cl, w1)
    b = ops.sdf_query_grid(axis, grid_cl, w2)
    assert a.shape == (1, Q ** 3, 1)
    assert torch.isfinite(a).all()
    torch.testing.assert_close(2.0 * a, b, atol=1e-6, rtol=1e-6)  # power-of-two scaling is exact in fp
