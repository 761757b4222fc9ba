"""Host-side mirror of the reference VQDIF module surface over libsfmi (HIP, gfx950).

Mirrors shapeformer/models/vqdif/vqdif.py:21-91 (`encode`, `encode_quant`, `quantize_cloud`,
`decode`, `decode_index`, `forward`) and the sub-module call contracts of enc.py / quantizer.py /
dec.py, with the same state-dict key names.  Every arithmetic step is a C-ABI
call into libsfmi.so; torch only owns device memory and the stream.  There is no CPU fallback.

Layout: all feature grids are channels-last (B,D,H,W,C) on device; reference-layout (B,C,D,H,W)
results are returned as zero-copy permuted views.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import weights as W


def _np(sd, k):
    v = sd[k]
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class _Conv:
    """One Conv3d (+ optional GroupNorm params that FOLLOW ('crg') or PRECEDE ('gcr') it)."""

    def __init__(self, sd, prefix, dev, ks, stride, pad):
        w = _np(sd, prefix + "conv.weight") if (prefix + "conv.weight") in sd else _np(sd, prefix + "weight")
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.ks, self.stride, self.pad = ks, stride, pad
        packed = np.empty(w.size, np.float32)
        L.check(L.lib().sfmi_conv_pack_weight(w.ctypes.data, self.cout, self.cin, ks, packed.ctypes.data), "conv_pack")
        self.w = torch.from_numpy(packed).to(dev)
        self.w_up = None     # sub-pixel weights (set for the convs that follow a nearest-x2 upsample)
        self.gamma = self.beta = self.bias = None
        if (prefix + "groupnorm.weight") in sd:
            self.gamma = torch.from_numpy(_np(sd, prefix + "groupnorm.weight")).to(dev)
            self.beta = torch.from_numpy(_np(sd, prefix + "groupnorm.bias")).to(dev)
        if (prefix + "bias") in sd:
            self.bias = torch.from_numpy(_np(sd, prefix + "bias")).to(dev)
        self._w_host = w

    def pack_subpixel(self, dev):
        """Pre-summed 2^3 weights of the 8 output parities for conv3(nearest_x2(x)) (csrc/conv3d.hip sfmi_conv3d_up2_cl_f32)."""
        assert self.ks == 3 and self.pad == 1 and self.stride == 1
        out = np.empty(64 * self.cout * self.cin, np.float32)
        L.check(L.lib().sfmi_conv_pack_weight_subpixel(np.ascontiguousarray(self._w_host).ctypes.data, self.cout, self.cin, out.ctypes.data),
                "conv_pack_subpixel")
        self.w_up = torch.from_numpy(out).to(dev)


class VQDIF:
    """Inference-side VQDIF (res16: d=128, 2 down/up steps; res32: d=64, 1 step)."""
    FUSE_DOWN0 = True     # False: the dense-grid route (mean grid + sfmi_conv3d_cl_f32), kept as the cross-check of the fused first conv

    G = 64
    GROUPS = 8
    EPS = 1e-5

    def __init__(self, state_dict=None, res=16, device="cuda:0", vocab_size=4096):
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.SfmiError("VQDIF needs a HIP device (no CPU fallback)")
        L.lib()
        self.res = res
        self.steps = 2 if res == 16 else 1
        self.d = 32 * 2 ** self.steps
        self.K = vocab_size
        sd = state_dict if state_dict is not None else W.make_state_dict(W.vqdif_spec(res))
        self.load_state_dict(sd)
        self._ws = {}

    # ------------------------------------------------------------------ weights
    def state_dict_np(self):
        """The (numpy, reference-layout) state dict the packed device weights were built from."""
        return self._sd

    def load_state_dict(self, sd):
        dev = self.dev
        lib = L.lib()
        self._sd = {k: np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}
        cat = lambda fmt, n=5: np.ascontiguousarray(np.stack([_np(sd, fmt.format(i)) for i in range(n)]))
        enc = np.empty(lib.sfmi_enc_pack_floats(), np.float32)
        a = [_np(sd, "encoder.fc_pos.weight"), _np(sd, "encoder.fc_pos.bias"), cat("encoder.blocks.{}.fc_0.weight"),
             cat("encoder.blocks.{}.fc_0.bias"), cat("encoder.blocks.{}.fc_1.weight"), cat("encoder.blocks.{}.fc_1.bias"),
             cat("encoder.blocks.{}.shortcut.weight"), _np(sd, "encoder.fc_c.weight"), _np(sd, "encoder.fc_c.bias"), enc]
        L.check(lib.sfmi_enc_pack_weights(*[x.ctypes.data for x in a]), "sfmi_enc_pack_weights")
        self.enc_w = torch.from_numpy(enc).to(dev)
        self.down = []
        for s in range(self.steps):
            self.down.append(_Conv(sd, f"encoder.downsampler.blocks.{2 * s}.", dev, 2, 2, 0))
            self.down.append(_Conv(sd, f"encoder.downsampler.blocks.{2 * s + 1}.", dev, 1, 1, 0))
        cb = _np(sd, "quantizer.embedding.weight")
        assert cb.shape == (self.K, self.d), cb.shape
        pk = np.empty(lib.sfmi_vq_pack_floats(self.K, self.d), np.float32)
        L.check(lib.sfmi_vq_pack_codebook(cb.ctypes.data, self.K, self.d, pk.ctypes.data), "sfmi_vq_pack_codebook")
        self.codebook = torch.from_numpy(cb).to(dev)
        self.codebook_packed = torch.from_numpy(pk).to(dev)
        u = "decoder.unet3d."
        self.unet = {}
        for name in ("encoders.0", "encoders.1", "encoders.2", "decoders.0", "decoders.1"):
            for sc in ("SingleConv1", "SingleConv2"):
                self.unet[f"{name}.{sc}"] = _Conv(sd, f"{u}{name}.basic_module.{sc}.", dev, 3, 1, 1)
        self.unet_final = _Conv(sd, u + "final_conv.", dev, 1, 1, 0)
        self.up = []
        for s in range(self.steps):
            self.up.append(_Conv(sd, f"decoder.upsampler.blocks.{3 * s + 1}.", dev, 3, 1, 1))
            self.up[-1].pack_subpixel(dev)
            self.up.append(_Conv(sd, f"decoder.upsampler.blocks.{3 * s + 2}.", dev, 3, 1, 1))
        from .ops import sdf_pack_weights
        self.sdf_w = torch.from_numpy(sdf_pack_weights(sd)).to(dev)

    # ------------------------------------------------------------------ primitives
    def _buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = torch.empty(shape, device=self.dev, dtype=dtype)
            self._ws[key] = t
        return t

    def _conv(self, x, cv, name, scale=None, shift=None, up=0, relu=True, bias=None):
        B, Di, Hi, Wi, Cin = x.shape
        assert Cin == cv.cin, (Cin, cv.cin)
        Do = ((Di << up) + 2 * cv.pad - cv.ks) // cv.stride + 1
        y = self._buf(name, (B, Do, Do, Do, cv.cout))
        if up and cv.w_up is not None:      # upsample + conv3 as 8 parity-wise 2^3 convolutions of the low-resolution grid
            L.check(L.lib().sfmi_conv3d_up2_cl_f32(L.ptr(x), L.ptr(cv.w_up), L.ptr(scale), L.ptr(shift), L.ptr(bias), L.ptr(y),
                                                   B, Di, Hi, Wi, Cin, cv.cout, int(relu), L.stream_ptr()), "sfmi_conv3d_up2_cl_f32")
            return y
        L.check(L.lib().sfmi_conv3d_cl_f32(L.ptr(x), L.ptr(cv.w), L.ptr(scale), L.ptr(shift), L.ptr(bias), L.ptr(y),
                                           B, Di, Hi, Wi, Cin, cv.cout, cv.ks, cv.stride, cv.pad, up, int(relu),
                                           L.stream_ptr()), "sfmi_conv3d_cl_f32")
        return y

    def _gn(self, x, gamma, beta, name):
        """scale/shift (B,C) such that GroupNorm8(x) == x*scale + shift."""
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        S = L.lib().sfmi_gn_splits(V)
        part = self._buf("gn_partial", (B * 64 * 1024 * 2,), torch.float64)
        assert B * S * C * 2 <= part.numel()
        sc, sh = self._buf(name + ".scale", (B, C)), self._buf(name + ".shift", (B, C))
        L.check(L.lib().sfmi_groupnorm_coeffs_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(sc), L.ptr(sh), L.ptr(part),
                                                  B, V, C, self.GROUPS, self.EPS, L.stream_ptr()), "sfmi_groupnorm_coeffs_f32")
        return sc, sh

    def _affine(self, x, sc, sh, name):
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        y = self._buf(name, tuple(x.shape))
        L.check(L.lib().sfmi_affine_cl_f32(L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(y), B, V, C, L.stream_ptr()), "sfmi_affine_cl_f32")
        return y

    # ------------------------------------------------------------------ encoder (a1-a8)
    def encode_cl(self, cloud):
        """cloud (B,T,3) in [-1,1] -> latent (B,R,R,R,d) channels-last, mask (B,R,R,R) uint8."""
        cloud = cloud.to(self.dev, torch.float32).contiguous()
        B, T, _ = cloud.shape
        R = self.res
        lib = L.lib()
        ws = self._buf("enc_ws", (lib.sfmi_enc_workspace_bytes(B, T),), torch.uint8)
        mask = self._buf("enc_mask", (B, R, R, R), torch.uint8)
        self.last_cell = self._buf("enc_cell", (B, T), torch.int32)
        d0 = self.down[0]
        first = 0
        if self.FUSE_DOWN0 and d0.ks == 2 and d0.stride == 2 and d0.cin == 32 and d0.cout == 64 and self.G == 64:
            # the first Downsampler convolution straight from the per-cell sums (csrc/encoder.hip:enc_down0_sparse_kernel): the
            # dense 64^3 x 32 mean grid (33.5 MB per shape) is neither zero-filled, scattered into, nor read back
            x = self._buf("down0", (B, self.G // 2, self.G // 2, self.G // 2, 64))
            L.check(lib.sfmi_encode_points_down_f32(L.ptr(cloud), L.ptr(self.enc_w), L.ptr(d0.w), L.ptr(x), L.ptr(mask),
                                                    L.ptr(self.last_cell), L.ptr(ws), B, T, R, 1, L.stream_ptr()), "sfmi_encode_points_down_f32")
            sc, sh = self._gn(x, d0.gamma, d0.beta, "down0")
            first = 1
        else:
            grid = self._buf("enc_grid", (B, self.G, self.G, self.G, 32))
            L.check(lib.sfmi_encode_points_f32(L.ptr(cloud), L.ptr(self.enc_w), L.ptr(grid), L.ptr(mask), L.ptr(self.last_cell),
                                               L.ptr(ws), B, T, R, L.stream_ptr()), "sfmi_encode_points_f32")
            x, sc, sh = grid, None, None
        for i, cv in enumerate(self.down):  # 'crg': conv -> ReLU -> GN (GN folded into the next consumer)
            if i < first:
                continue
            x = self._conv(x, cv, f"down{i}", sc, sh, relu=True)
            sc, sh = self._gn(x, cv.gamma, cv.beta, f"down{i}")
        latent = self._affine(x, sc, sh, "latent")
        return latent, mask

    def encode(self, Xbd):
        """vqdif.py:35-37 -> (grid_feat (B,d,R,R,R) view, grid_mask (B,R,R,R) bool)."""
        lat, mask = self.encode_cl(Xbd)
        return lat.permute(0, 4, 1, 2, 3).clone(), mask.bool()      # fresh tensors: the *_cl / *_dev paths return workspace views

    # ------------------------------------------------------------------ quantizer (a9, a10)
    def quantize_cl(self, latent):
        N = latent.numel() // self.d
        idx = self._buf("vq_idx", (N,), torch.int32)
        L.check(L.lib().sfmi_vq_argmin_f32(L.ptr(latent), L.ptr(self.codebook_packed), L.ptr(idx), None, N, self.K, self.d,
                                           L.stream_ptr()), "sfmi_vq_argmin_f32")
        return idx.view(latent.shape[:-1])

    def get_code_cl(self, ind):
        """quantizer.py:19-30 -> channels-last (B,R,R,R,d)."""
        ind32 = ind.to(self.dev, torch.int32).contiguous()
        N = ind32.numel()
        out = self._buf("vq_code", tuple(ind32.shape) + (self.d,))
        L.check(L.lib().sfmi_vq_gather_f32(L.ptr(self.codebook), L.ptr(ind32), L.ptr(out), N, self.d, L.stream_ptr()), "sfmi_vq_gather_f32")
        return out

    def mode_of(self, idx, name="mode", rows=1):
        hist = self._buf("hist", (rows * (self.K + 1),), torch.int32)
        mode = self._buf(name, (rows,), torch.int32)
        L.check(L.lib().sfmi_mode_i32(L.ptr(idx), idx.numel(), self.K + 1, rows, L.ptr(hist), L.ptr(mode), L.stream_ptr()), "sfmi_mode_i32")
        return mode

    def quantize_cloud_dev(self, cloud, per_shape_mode=False):
        """Device-resident quantize_cloud: (quant_ind int32 (B,R,R,R), mode int32, raw idx, mask u8, latent).

        per_shape_mode=False reproduces the reference on a batch (ONE mode over the whole batch, vqdif.py:53);
        True gives every shape its own empty code = the reference run shape-by-shape at batch size 1, which is
        how its inference drivers call it (shapeformer.py:227 asserts batch_size == 1)."""
        latent, mask = self.encode_cl(cloud)
        raw = self.quantize_cl(latent)
        rows = raw.shape[0] if per_shape_mode else 1
        mode = self.mode_of(raw, rows=rows)
        q = self._buf("quant_ind", tuple(raw.shape), torch.int32)
        L.check(L.lib().sfmi_apply_mask_i32(L.ptr(raw), L.ptr(mask), L.ptr(mode), L.ptr(q), raw.numel(), rows, L.stream_ptr()), "sfmi_apply_mask_i32")
        return q, mode, raw, mask, latent

    def quantize_cloud(self, cloud):
        """vqdif.py:50-58 -> (quant_ind (B,R,R,R) int64, mode, dict(quant_ind, grid_mask, quant_feat...))."""
        q, mode, raw, mask, latent = self.quantize_cloud_dev(cloud)
        code = self.get_code_cl(raw)
        # encode_quant's dict (vqdif.py:39-48); fresh tensors (the *_dev path returns views of a reused workspace)
        enc = dict(quant_feat=code.permute(0, 4, 1, 2, 3).clone(), quant_ind=raw.long(), quant_diff=((latent - code) ** 2).mean(),
                   grid_mask=mask.bool(), grid_feat=latent.permute(0, 4, 1, 2, 3).clone())
        return q.long(), mode.long()[0], enc

    # ------------------------------------------------------------------ decoder grid (a21, a22)
    def _single_gcr(self, x, cv, name):
        sc, sh = self._gn(x, cv.gamma, cv.beta, name)
        return self._conv(x, cv, name, sc, sh, relu=True)

    def _double(self, x, name):
        x = self._single_gcr(x, self.unet[name + ".SingleConv1"], name + ".c1")
        return self._single_gcr(x, self.unet[name + ".SingleConv2"], name + ".c2")

    def _pool(self, x, name):
        B, D, _, _, C = x.shape
        y = self._buf(name, (B, D // 2, D // 2, D // 2, C))
        L.check(L.lib().sfmi_maxpool2_cl_f32(L.ptr(x), L.ptr(y), B, D // 2, D // 2, D // 2, C, L.stream_ptr()), "sfmi_maxpool2_cl_f32")
        return y

    def _upcat(self, skip, low, name):
        B, D, _, _, Cs = skip.shape
        Cu = low.shape[-1]
        y = self._buf(name, (B, D, D, D, Cs + Cu))
        L.check(L.lib().sfmi_upcat_cl_f32(L.ptr(skip), L.ptr(low), L.ptr(y), B, D, D, D, Cs, Cu, L.stream_ptr()), "sfmi_upcat_cl_f32")
        return y

    def decoder_grid_cl(self, code_cl, final_affine=True):
        """dec.py:75-83: UNet3D + Upsampler -> (B,64,64,64,32) channels-last."""
        e0 = self._double(code_cl, "encoders.0")
        e1 = self._double(self._pool(e0, "pool0"), "encoders.1")
        e2 = self._double(self._pool(e1, "pool1"), "encoders.2")
        y = self._double(self._upcat(e1, e2, "cat0"), "decoders.0")
        y = self._double(self._upcat(e0, y, "cat1"), "decoders.1")
        x = self._conv(y, self.unet_final, "unet_out", relu=False, bias=self.unet_final.bias)
        sc = sh = None
        for i, cv in enumerate(self.up):  # nearest x2 folded into the first conv of each step
            x = self._conv(x, cv, f"up{i}", sc, sh, up=1 if i % 2 == 0 else 0, relu=True)
            sc, sh = self._gn(x, cv.gamma, cv.beta, f"up{i}")
        if final_affine:
            return self._affine(x, sc, sh, "dec_grid")
        return x, sc, sh

    # ------------------------------------------------------------------ a20, a23
    def decode_index(self, code_ind, Xtg=None, grid_Q=None, sigmoid=False):
        """vqdif.py:60-76. Xtg (B,N,3) arbitrary points, or grid_Q=Q for the makeGrid 'ij' Q^3 lattice."""
        from . import ops
        grid = self.decoder_grid_cl(self.get_code_cl(code_ind))
        if grid_Q is not None:
            axis = torch.from_numpy(np.linspace(-1.0, 1.0, grid_Q).astype(np.float32)).to(self.dev)
            return dict(logits=ops.sdf_query_grid(axis, grid, self.sdf_w, sigmoid=sigmoid))
        return dict(logits=ops.sdf_query(Xtg.to(self.dev, torch.float32), grid, self.sdf_w, sigmoid=sigmoid))

    def decode(self, grid_feat, Xtg):
        """vqdif.py:60-72 with a (B,d,R,R,R) feature grid."""
        from . import ops
        code = grid_feat.permute(0, 2, 3, 4, 1).contiguous().to(self.dev, torch.float32)
        grid = self.decoder_grid_cl(code)
        return dict(logits=ops.sdf_query(Xtg.to(self.dev, torch.float32), grid, self.sdf_w))

    def forward(self, Xbd, Xtg):
        """vqdif.py:78-91 (eval): encode -> quantize -> decode."""
        latent, mask = self.encode_cl(Xbd)
        raw = self.quantize_cl(latent)
        out = self.decode_index(raw, Xtg)
        q = self.get_code_cl(raw)
        return dict(logits=out["logits"], quant_feat=q.permute(0, 4, 1, 2, 3).clone(), quant_ind=raw.long(),
                    quant_diff=((latent - q) ** 2).mean(), grid_mask=mask.bool())

    __call__ = forward
