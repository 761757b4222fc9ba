"""The local-pool encoder's five per-point stages in one launch (csrc/encoder.hip enc_fused_kernel, knob enc_fused = 1, the default)
against the one-launch-per-stage form (enc_fused = 0): BIT-IDENTICAL latent grids / first-Downsampler outputs, masks and cell ids -
per point the arithmetic is the same instruction sequence, the max pool is an integer max and the mean a fixed-point sum
(enc.py:95-140).  Cases: the bench's synthetic clouds; a point count that is no multiple of the workgroups' 384 positions; cells
that hold exactly 128 points (the longest run the fused form takes); a batch in which ONE shape has a cell of 129 points (the scan
raises that shape's flag: the fused kernel skips it and the staged kernels behind it take exactly that shape - the same launch
sequence, no host decision); a cloud that is one cell."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, L, cloud, enc_w, w_down, knob, dev, down):
    B, T, _ = cloud.shape
    L.check(lib.sfmi_tune_set(b"enc_fused", knob), "tune")
    try:
        ws = torch.empty(lib.sfmi_enc_workspace_bytes(B, T), device=dev, dtype=torch.uint8)
        mask = torch.empty(B, 16, 16, 16, device=dev, dtype=torch.uint8)
        cell = torch.empty(B, T, device=dev, dtype=torch.int32)
        if down:
            y = torch.empty(B, 32, 32, 32, 64, device=dev)
            L.check(lib.sfmi_encode_points_down_f32(L.ptr(cloud), L.ptr(enc_w), L.ptr(w_down), L.ptr(y), L.ptr(mask), L.ptr(cell), L.ptr(ws),
                                                    B, T, 16, 1, L.stream_ptr()), "encode_down")
        else:
            y = torch.empty(B, 64, 64, 64, 32, device=dev)
            L.check(lib.sfmi_encode_points_f32(L.ptr(cloud), L.ptr(enc_w), L.ptr(y), L.ptr(mask), L.ptr(cell), L.ptr(ws), B, T, 16,
                                               L.stream_ptr()), "encode")
        torch.cuda.synchronize()
        return y, mask, cell
    finally:
        L.check(lib.sfmi_tune_set(b"enc_fused", 1), "tune")


def _clouds():
    from shapeformer_amd import synthetic
    g = np.random.default_rng(11)
    out = {"synthetic 3 x 16384": synthetic.make_batch(5, 3, n_partial=16384)["Xct"].astype(np.float32),
           "synthetic 2 x 1000": synthetic.make_batch(6, 2, n_partial=1000)["Xct"].astype(np.float32)}

    def clustered(T, sizes_per_shape):
        c = (g.random((len(sizes_per_shape), T, 3), dtype=np.float32) - 0.5) * 1.2            # the bulk: [-0.6, 0.6]^3
        for b, sizes in enumerate(sizes_per_shape):
            at = 0
            for k, n in enumerate(sizes):                                     # n coincident points in a cell far from the bulk
                c[b, at:at + n] = np.float32([0.9 - 0.05 * k, 0.9, 0.9]) + (g.random((n, 3), dtype=np.float32) - 0.5) * 1e-3
                at += n
            c[b] = c[b, g.permutation(T)]
        return c
    out["cells of 128 / 127 / 65 points"] = clustered(4000, [[128, 127, 65, 128], [33, 128]])
    out["one shape with a cell of 129 points"] = clustered(4000, [[20, 30], [129, 10], [128]])
    out["one cell"] = np.float32([0.3, -0.2, 0.1]) + (g.random((2, 500, 3), dtype=np.float32) - 0.5) * 1e-3
    return out


def test_fused_stage_kernel_is_bit_identical_to_the_staged_form(dev):
    from shapeformer_amd import _lib as L
    from shapeformer_amd.vqdif import VQDIF
    lib = L.lib()
    vq = VQDIF(res=16, device=dev)
    assert lib.sfmi_tune_get(b"enc_fused") == 1
    for name, c in _clouds().items():
        cloud = torch.from_numpy(np.ascontiguousarray(c)).to(dev)
        for down in (True, False):
            y1, m1, c1 = _run(lib, L, cloud, vq.enc_w, vq.down[0].w, 1, dev, down)
            y0, m0, c0 = _run(lib, L, cloud, vq.enc_w, vq.down[0].w, 0, dev, down)
            assert torch.equal(c1, c0) and torch.equal(m1, m0), name
            assert torch.equal(y1, y0), (name, down, float((y1 - y0).abs().max()))
            assert float(y1.abs().sum()) > 0, name
