"""Thin Python wrappers over the C ABI (one function per exported kernel launcher).

Tensors are torch CUDA tensors used purely as device-memory handles; every call
goes through libsfmi.so (shapeformer_amd/_lib.py).  No eager fallbacks.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.SfmiError("libsfmi kernels need CUDA/HIP device tensors (no CPU fallback)")


def _c(t, dtype):
    assert t.dtype == dtype, (t.dtype, dtype)
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------- SDF query
def sdf_pack_weights(sd, prefix="decoder.") -> np.ndarray:
    """Pack LocalDecoder MLP tensors (reference key names) into the kernel's fragment order."""
    g = lambda k: np.ascontiguousarray(np.asarray(sd[prefix + k], dtype=np.float32))
    cat = lambda fmt: np.ascontiguousarray(np.stack([g(fmt.format(i)) for i in range(5)]))
    out = np.empty(L.lib().sfmi_sdf_pack_floats(), np.float32)
    arrs = [g("fc_p.weight"), g("fc_p.bias"), cat("fc_c.{}.weight"), cat("fc_c.{}.bias"),
            cat("blocks.{}.fc_0.weight"), cat("blocks.{}.fc_0.bias"), cat("blocks.{}.fc_1.weight"),
            cat("blocks.{}.fc_1.bias"), g("fc_out.weight"), g("fc_out.bias"), out]
    L.check(L.lib().sfmi_sdf_pack_weights(*[a.ctypes.data for a in arrs]), "sfmi_sdf_pack_weights")
    return out


def sdf_query(xyz, grid_cl, wpack, sigmoid=False, out=None):
    """xyz (B,N,3) in [-1,1]; grid_cl (B,G,G,G,32) channels-last; -> (B,N,1) logits."""
    _chk_cuda(xyz, grid_cl, wpack)
    xyz, grid_cl = _c(xyz, torch.float32), _c(grid_cl, torch.float32)
    B, N, _ = xyz.shape
    G = grid_cl.shape[1]
    assert grid_cl.shape == (B, G, G, G, 32)
    if out is None:
        out = torch.empty(B, N, 1, device=xyz.device, dtype=torch.float32)
    L.check(L.lib().sfmi_sdf_query_f32(L.ptr(xyz), L.ptr(grid_cl), L.ptr(wpack), L.ptr(out), B, N, G,
                                       int(sigmoid), L.stream_ptr()), "sfmi_sdf_query_f32")
    return out


def sigmoid(x, out=None):
    """nputil.sigmoid of a device logits tensor (in-tree kernel: the SDF query's own epilogue expression)."""
    _chk_cuda(x)
    x = _c(x, torch.float32)
    if x.data_ptr() % 16:          # a contiguous view that starts inside a 16-byte line: the kernel reads float4
        x = x.clone()
    out = torch.empty_like(x) if out is None else out
    assert out.data_ptr() % 16 == 0 and out.is_contiguous()
    L.check(L.lib().sfmi_sigmoid_f32(L.ptr(x), L.ptr(out), x.numel(), L.stream_ptr()), "sfmi_sigmoid_f32")
    return out


def sdf_query_grid(axis, grid_cl, wpack, sigmoid=False, out=None, x_range=None, affine=None):
    """Structured Q^3 'ij' query grid from a Q-entry f32 axis table -> (B,Q^3,1); x_range = (x0, x1): only the planes x0 <= ix < x1 of
    the slowest lattice index -> (B,(x1-x0) Q^2,1), the same values the whole-lattice call computes for them.  affine = (scale, shift)
    (B,32) each: grid_cl is the decoder grid BEFORE its last GroupNorm, whose affine the kernel applies to the interpolated features."""
    _chk_cuda(axis, grid_cl, wpack)
    axis, grid_cl = _c(axis, torch.float32), _c(grid_cl, torch.float32)
    Q = axis.numel()
    x0, x1 = (0, Q) if x_range is None else (int(x_range[0]), int(x_range[1]))
    B, G = grid_cl.shape[0], grid_cl.shape[1]
    if out is None:
        out = torch.empty(B, (x1 - x0) * Q * Q, 1, device=axis.device, dtype=torch.float32)
    sc, sh = (None, None) if affine is None else (_c(affine[0], torch.float32), _c(affine[1], torch.float32))
    assert affine is None or (sc.shape == (B, 32) and sh.shape == (B, 32))
    L.check(L.lib().sfmi_sdf_query_grid_aff_f32(L.ptr(axis), Q, x0, x1, L.ptr(grid_cl), L.ptr(sc), L.ptr(sh), L.ptr(wpack), L.ptr(out), B, G,
                                                int(sigmoid), L.stream_ptr()), "sfmi_sdf_query_grid_aff_f32")
    return out
