"""Iso-surface extraction on the GPU — host side of csrc/mcubes.hip.

`array2mesh` mirrors xgutils/geoutil.py:175-233 for dim == 3 (no gaussian filter, no decimation, cart_coord=True):
marching cubes at `thresh`, vertices mapped onto the bounding box of the query coordinates.  The reference calls
PyMCubes on the CPU after copying the 128^3 occupancy to the host; here the grid never leaves HBM and the result is an
indexed mesh (shared vertices), in a deterministic order.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def marching_cubes_dev(occ: torch.Tensor, thresh: float = 0.5, bbox=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))):
    """occ (B,Q,Q,Q) f32 on the HIP device (axes = grid axes 0,1,2 == x,y,z of nputil.makeGrid 'ij').

    -> verts (V,3) f32, faces (T,3) int32 (indices local to each shape), voff (B+1,), toff (B+1,) host int arrays."""
    if occ.device.type != "cuda":
        raise L.SfmiError("marching_cubes_dev needs a HIP device tensor (no CPU fallback)")
    lib = L.lib()
    occ = occ.contiguous().float()
    B, Q = occ.shape[0], occ.shape[1]
    assert occ.shape == (B, Q, Q, Q)
    ws = torch.empty(lib.sfmi_mc_workspace_bytes(B, Q), device=occ.device, dtype=torch.uint8)
    offs = torch.zeros(2 * (B + 1), device=occ.device, dtype=torch.int32)
    L.check(lib.sfmi_mc_count_f32(L.ptr(occ), float(thresh), B, Q, L.ptr(ws), L.ptr(offs), L.stream_ptr()), "sfmi_mc_count_f32")
    o = offs.cpu().numpy()            # the one host sync: output sizes
    voff, toff = o[:B + 1].copy(), o[B + 1:].copy()
    verts = torch.empty(max(int(voff[-1]), 1), 3, device=occ.device, dtype=torch.float32)
    faces = torch.empty(max(int(toff[-1]), 1), 3, device=occ.device, dtype=torch.int32)
    lo, hi = bbox
    L.check(lib.sfmi_mc_emit_f32(L.ptr(occ), float(thresh), B, Q, L.ptr(ws), L.ptr(offs), float(lo[0]), float(lo[1]), float(lo[2]),
                                 float(hi[0]), float(hi[1]), float(hi[2]), L.ptr(verts), L.ptr(faces), L.stream_ptr()),
            "sfmi_mc_emit_f32")
    return verts[:int(voff[-1])], faces[:int(toff[-1])], voff, toff


def array2mesh(occupancy, thresh=0.5, coords=None, bbox=((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0)), device="cuda:0"):
    """geoutil.array2mesh(array, thresh, dim=3, coords=...) for ONE flattened or cubic grid -> (verts (V,3) float64,
    faces (T,3) int) numpy, like the reference returns (`verts*(bbmax-bbmin)+bbmin`, faces.astype(int))."""
    a = torch.as_tensor(occupancy, dtype=torch.float32)
    if a.dim() != 3:
        Q = round(a.numel() ** (1.0 / 3))
        assert Q ** 3 == a.numel(), "array2NDCube: not a cube"
        a = a.reshape(Q, Q, Q)
    if coords is not None:
        c = np.asarray(coords).reshape(-1, 3)
        bbox = (c.min(0), c.max(0))                # nputil.arrayBBox
    v, f, _, _ = marching_cubes_dev(a[None].to(device), thresh, bbox)
    return v.cpu().numpy().astype(np.float64), f.cpu().numpy().astype(int)
