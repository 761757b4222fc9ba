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


class _Workspace:
    """Named device buffers, reused from call to call (no allocation on the steady-state path)."""

    def __init__(self, dev):
        self.dev, self.bufs = dev, {}

    def buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = torch.empty(shape, device=self.dev, dtype=dtype)
            self.bufs[key] = t
        return t


def _dev_of(device):
    dev = torch.device(device if device is not None else "cuda:0")
    if dev.type != "cuda":
        raise L.SfmiError("the VQDIF modules need a HIP device (no CPU fallback)")
    return dev


def _twice(p, dev):
    """[-.5,.5]^3 module coordinates -> the [-1,1]^3 the kernels take (they halve again: vqdif.py:36,71).  p + p on the device:
    exact in f32."""
    p = torch.as_tensor(p).to(dev, torch.float32).contiguous()
    out = torch.empty_like(p)
    L.check(L.lib().sfmi_add_f32(L.ptr(p), L.ptr(p), L.ptr(out), p.numel(), L.stream_ptr()), "sfmi_add_f32")
    return out


def _np_sd(sd):
    return {k: np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}


def _limit(cond, what):
    """Hyper-parameters outside what the HIP kernels are built for: say which, and what the limit is."""
    if not cond:
        raise ValueError(f"shapeformer_amd: unsupported hyper-parameter: {what}")


def _sub_sd(state_dict, prefix, res):
    """Sub-module weights: `state_dict` with or without the `prefix` ("encoder." ...); None -> the hash-generated
    weights of the full VQDIF of that latent resolution (the same values VQDIF(res=...) builds)."""
    if state_dict is None:
        state_dict = W.make_state_dict(W.vqdif_spec(res))
    if any(k.startswith(prefix) for k in state_dict):
        return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
    return dict(state_dict)


class _GridOps:
    """Convolution / GroupNorm helpers shared by the encoder's Downsampler and the decoder's UNet3D + Upsampler."""
    GROUPS = 8
    EPS = 1e-5

    def _buf(self, name, shape, dtype=torch.float32):
        return self.ws.buf(name, shape, dtype)

    def _gn_partial_buf(self, B):
        return self._buf("gn_partial", (B * 64 * 1024 * 2,), torch.float64)

    def _conv(self, x, cv, name, scale=None, shift=None, up=0, relu=True, bias=None, stats=False):
        """stats=True -> (y, S): the launch also leaves y's GroupNorm statistics as S per-shape partials in the `gn_partial` buffer
        (csrc/conv3d.hip ST instances; consumed by `_gn(..., partial_S=S)`), or S = None where this geometry has no such instance."""
        import ctypes as C
        B, Di, Hi, Wi, Cin = x.shape
        assert Cin == cv.cin, (Cin, cv.cin)
        Do = ((Di << up) + 2 * cv.pad - cv.ks) // cv.stride + 1
        y = self._buf(name, (B, Do, Do, Do, cv.cout))
        S = C.c_int(0)
        part = self._gn_partial_buf(B) if stats else None
        if up and cv.w_up is not None:      # upsample + conv3 as 8 parity-wise 2^3 convolutions of the low-resolution grid
            rc = L.lib().sfmi_conv3d_up2_cl_stats_f32(L.ptr(x), L.ptr(cv.w_up), L.ptr(scale), L.ptr(shift), L.ptr(bias), L.ptr(y),
                                                      B, Di, Hi, Wi, Cin, cv.cout, int(relu), L.ptr(part), C.byref(S) if stats else None,
                                                      L.stream_ptr()) if stats else L.SFMI_EINVAL
            if rc == L.SFMI_EINVAL:         # no statistics instance for this geometry (or none asked for): the plain launches
                S = None
                rc = L.lib().sfmi_conv3d_up2_cl_f32(L.ptr(x), L.ptr(cv.w_up), L.ptr(scale), L.ptr(shift), L.ptr(bias), L.ptr(y),
                                                    B, Di, Hi, Wi, Cin, cv.cout, int(relu), L.stream_ptr())
            L.check(rc, "sfmi_conv3d_up2_cl_f32")
        else:
            rc = L.lib().sfmi_conv3d_cl_stats_f32(L.ptr(x), L.ptr(cv.w), L.ptr(scale), L.ptr(shift), L.ptr(bias), L.ptr(y),
                                                  B, Di, Hi, Wi, Cin, cv.cout, cv.ks, cv.stride, cv.pad, up, int(relu), L.ptr(part),
                                                  C.byref(S) if stats else None, L.stream_ptr()) if stats else L.SFMI_EINVAL
            if rc == L.SFMI_EINVAL:
                S = None
                rc = L.lib().sfmi_conv3d_cl_f32(L.ptr(x), L.ptr(cv.w), L.ptr(scale), L.ptr(shift), L.ptr(bias), L.ptr(y),
                                                B, Di, Hi, Wi, Cin, cv.cout, cv.ks, cv.stride, cv.pad, up, int(relu), L.stream_ptr())
            L.check(rc, "sfmi_conv3d_cl_f32")
        if not stats:
            return y
        return y, (None if S is None else int(S.value))

    def _gn(self, x, gamma, beta, name, partial_S=None):
        """scale/shift (B,C) such that GroupNorm8(x) == x*scale + shift.  partial_S: the convolution that wrote x left its statistics in
        the `gn_partial` buffer (`_conv(..., stats=True)`): only the coefficient launch runs."""
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        part = self._gn_partial_buf(B)
        sc, sh = self._buf(name + ".scale", (B, C)), self._buf(name + ".shift", (B, C))
        if partial_S is not None:
            assert B * partial_S * C * 2 <= part.numel()
            L.check(L.lib().sfmi_groupnorm_coeffs_partial_f32(L.ptr(part), L.ptr(gamma), L.ptr(beta), L.ptr(sc), L.ptr(sh), B, V, C, int(partial_S),
                                                              self.GROUPS, self.EPS, L.stream_ptr()), "sfmi_groupnorm_coeffs_partial_f32")
            return sc, sh
        S = L.lib().sfmi_gn_splits(V)
        assert B * S * C * 2 <= part.numel()
        L.check(L.lib().sfmi_groupnorm_coeffs_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(sc), L.ptr(sh), L.ptr(part),
                                                  B, V, C, self.GROUPS, self.EPS, L.stream_ptr()), "sfmi_groupnorm_coeffs_f32")
        return sc, sh

    def _affine(self, x, sc, sh, name):
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        y = self._buf(name, tuple(x.shape))
        L.check(L.lib().sfmi_affine_cl_f32(L.ptr(x), L.ptr(sc), L.ptr(sh), L.ptr(y), B, V, C, L.stream_ptr()), "sfmi_affine_cl_f32")
        return y


class LocalPoolPointnet(_GridOps):
    """`shapeformer.models.vqdif.enc.LocalPoolPointnet` (enc.py:11-140): ctor kwargs, state-dict keys (`fc_pos.*`,
    `blocks.{0-4}.*`, `fc_c.*`, `downsampler.blocks.*`) and call contract `encoder(p) -> (fea (B,k*C,R,R,R), mask (B,R,R,R) bool)`
    with p = Xbd / 2 in [-.5,.5]^3 (vqdif.py:35-37).  One C-ABI call runs the point MLP, the four local max-pools, the
    per-cell mean and the first Downsampler convolution (csrc/encoder.hip); the remaining Downsampler convolutions and the
    GroupNorms are csrc/conv3d.hip.  Built for the shipped hyper-parameters; anything else raises ValueError naming the limit."""
    FUSE_DOWN0 = True     # False: the dense-grid route (mean grid + sfmi_conv3d_cl_f32), kept as the cross-check of the fused first conv
    G = 64

    def __init__(self, c_dim=128, dim=3, hidden_dim=128, scatter_type="max", downsampler=False, downsampler_kwargs=None,
                 c2i_order="original", grid_resolution=None, plane_type="grid", padding=0.1, n_blocks=5, state_dict=None,
                 device=None, workspace=None):
        _limit(c_dim == 32 and hidden_dim == 32, f"LocalPoolPointnet(c_dim={c_dim}, hidden_dim={hidden_dim}): the point-MLP kernel is built for 32 / 32")
        _limit(dim == 3 and n_blocks == 5, f"LocalPoolPointnet(dim={dim}, n_blocks={n_blocks}): 3-D points, 5 ResnetBlockFC stages")
        _limit(scatter_type == "max", f"LocalPoolPointnet(scatter_type={scatter_type!r}): only 'max' local pooling")
        _limit(c2i_order == "original", f"LocalPoolPointnet(c2i_order={c2i_order!r}): only 'original' (x fastest)")
        _limit(grid_resolution == 64, f"LocalPoolPointnet(grid_resolution={grid_resolution}): the cell sort is built for the 64^3 feature grid")
        _limit("grid" in plane_type, f"LocalPoolPointnet(plane_type={plane_type!r}): only the 3-D 'grid' feature volume")
        _limit(abs(padding - 0.1) < 1e-12, f"LocalPoolPointnet(padding={padding}): the coordinate normalisation is built for 0.1")
        dk = downsampler_kwargs or {}
        _limit(bool(downsampler) and dk.get("in_channels") == 32 and dk.get("downsample_steps") in (1, 2),
               f"LocalPoolPointnet(downsampler={downsampler}, downsampler_kwargs={dk}): a Downsampler of 1 or 2 steps on 32 channels")
        self.c_dim, self.hidden_dim, self.reso_grid, self.plane_type, self.padding = c_dim, hidden_dim, grid_resolution, plane_type, padding
        self.steps = int(dk["downsample_steps"])
        self.res = self.G >> self.steps
        self.dev = _dev_of(device)
        L.lib()
        self.ws = workspace or _Workspace(self.dev)
        self.load_state_dict(_sub_sd(state_dict, "encoder.", self.res))

    def load_state_dict(self, sd):
        lib, dev = L.lib(), self.dev
        self._sd = _np_sd(sd)
        cat = lambda fmt, n=5: np.ascontiguousarray(np.stack([_np(sd, fmt.format(i)) for i in range(n)]))
        enc = np.empty(lib.sfmi_enc_pack_floats(), np.float32)
        a = [_np(sd, "fc_pos.weight"), _np(sd, "fc_pos.bias"), cat("blocks.{}.fc_0.weight"),
             cat("blocks.{}.fc_0.bias"), cat("blocks.{}.fc_1.weight"), cat("blocks.{}.fc_1.bias"),
             cat("blocks.{}.shortcut.weight"), _np(sd, "fc_c.weight"), _np(sd, "fc_c.bias"), enc]
        L.check(lib.sfmi_enc_pack_weights(*[x.ctypes.data for x in a]), "sfmi_enc_pack_weights")
        self.enc_w = torch.from_numpy(enc).to(dev)
        self.down = []
        for s in range(self.steps):
            self.down.append(_Conv(sd, f"downsampler.blocks.{2 * s}.", dev, 2, 2, 0))
            self.down.append(_Conv(sd, f"downsampler.blocks.{2 * s + 1}.", dev, 1, 1, 0))

    def encode_cl(self, cloud):
        """cloud (B,T,3) in [-1,1] -> latent (B,R,R,R,d) channels-last, mask (B,R,R,R) uint8 (workspace views)."""
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

    def forward(self, p):
        """enc.py:115-140: p (B,T,3) in [-.5,.5]^3 -> (fea (B,k*C,R,R,R), mask (B,R,R,R) bool), fresh tensors.  The kernels take
        the cloud in [-1,1] and halve it themselves; the doubling here (p + p) is exact in f32."""
        lat, mask = self.encode_cl(_twice(p, self.dev))
        return lat.permute(0, 4, 1, 2, 3).clone(), mask.bool()

    __call__ = forward


class Quantizer:
    """`shapeformer.models.vqdif.quantizer.Quantizer` (quantizer.py:6-89): `Quantizer(vocab_size, n_embd, gamma, x_dim)`,
    state-dict keys `embedding.weight`, `N`, `z_avg`; `quantizer(x (B,d,R,R,R)) -> (quant_feat, quant_feat_st, idx (B,R,R,R)
    int64, quant_diff)`, `get_code(idx) -> (B,d,R,R,R)`.  The nearest-code search is csrc/vq_argmin.hip (MFMA distance tiles +
    wavefront argmin, lowest index on ties = torch.max on CPU); this is the EVAL-mode module - the EMA codebook update of
    training mode (quantizer.py:68-83) lives in train_vqdif.VQDIFTrainer."""
    training = False

    def __init__(self, vocab_size, n_embd, gamma=0.99, x_dim=3, state_dict=None, device=None, workspace=None):
        _limit(n_embd in (64, 128), f"Quantizer(n_embd={n_embd}): the distance kernel is built for 64 / 128 channels")
        _limit(vocab_size % 32 == 0 and vocab_size > 0, f"Quantizer(vocab_size={vocab_size}): a positive multiple of 32 codes")
        _limit(x_dim == 3, f"Quantizer(x_dim={x_dim}): 3-D latent grids")
        self.vocab_size, self.n_embd, self.gamma, self.x_dim = vocab_size, n_embd, gamma, x_dim
        self.K, self.d = vocab_size, n_embd
        self.dev = _dev_of(device)
        L.lib()
        self.ws = workspace or _Workspace(self.dev)
        self.load_state_dict(_sub_sd(state_dict, "quantizer.", 16 if n_embd == 128 else 32))

    def _buf(self, name, shape, dtype=torch.float32):
        return self.ws.buf(name, shape, dtype)

    def load_state_dict(self, sd):
        lib = L.lib()
        self._sd = _np_sd(sd)
        cb = _np(sd, "embedding.weight")
        _limit(cb.shape == (self.K, self.d), f"Quantizer: embedding.weight {cb.shape} != (vocab_size, n_embd) = {(self.K, self.d)}")
        pk = np.empty(lib.sfmi_vq_pack_floats(self.K, self.d), np.float32)
        L.check(lib.sfmi_vq_pack_codebook(cb.ctypes.data, self.K, self.d, pk.ctypes.data), "sfmi_vq_pack_codebook")
        self.codebook = torch.from_numpy(cb).to(self.dev)
        self.codebook_packed = torch.from_numpy(pk).to(self.dev)

    def quantize_cl(self, latent):
        """channels-last latent (...,d) -> nearest-code indices (...) int32 (workspace view)."""
        N = latent.numel() // self.d
        idx = self._buf("vq_idx", (N,), torch.int32)
        L.check(L.lib().sfmi_vq_argmin_f32(L.ptr(latent), L.ptr(self.codebook_packed), L.ptr(idx), None, N, self.K, self.d,
                                           L.stream_ptr()), "sfmi_vq_argmin_f32")
        return idx.view(latent.shape[:-1])

    def get_code_cl(self, ind):
        """quantizer.py:19-30 -> channels-last (B,R,R,R,d) (workspace view)."""
        ind32 = ind.to(self.dev, torch.int32).contiguous()
        N = ind32.numel()
        out = self._buf("vq_code", tuple(ind32.shape) + (self.d,))
        L.check(L.lib().sfmi_vq_gather_f32(L.ptr(self.codebook), L.ptr(ind32), L.ptr(out), N, self.d, L.stream_ptr()), "sfmi_vq_gather_f32")
        return out

    def get_code(self, ind, bchw=True):
        """quantizer.py:19-30: code vectors of an index grid, (B,d,R,R,R) (bchw) or (B,R,R,R,d); a fresh tensor."""
        code = self.get_code_cl(torch.as_tensor(ind))
        return code.permute(0, 4, 1, 2, 3).contiguous() if bchw else code.clone()

    def forward(self, grid_feat):
        """quantizer.py:31-89 (eval): (B,d,R,R,R) -> (quant_feat, quant_feat_st, encoding_indices int64, quant_diff)."""
        if self.training:
            raise L.SfmiError("Quantizer: the training-mode EMA update runs in train_vqdif.VQDIFTrainer; this module is eval-mode")
        x = torch.as_tensor(grid_feat).to(self.dev, torch.float32)
        _limit(x.dim() == 5 and x.shape[1] == self.d, f"Quantizer input {tuple(x.shape)}: a (B,{self.d},R,R,R) grid")
        lat = x.permute(0, 2, 3, 4, 1).contiguous()
        idx = self.quantize_cl(lat)
        code = self.get_code_cl(idx)
        q = code.permute(0, 4, 1, 2, 3).contiguous()
        return q, q.clone(), idx.long(), ((lat - code) ** 2).mean()      # eval: quant_feat_st == quant_feat numerically (:86-87)

    __call__ = forward


class LocalDecoder(_GridOps):
    """`shapeformer.models.vqdif.dec.LocalDecoder` (dec.py:10-100): ctor kwargs, state-dict keys (`unet3d.*`, `upsampler.*`,
    `fc_c.{0-4}.*`, `fc_p.*`, `blocks.{0-4}.*`, `fc_out.*`), `decoder(p (B,N,3) in [-.5,.5]^3, c_grid (B,d,R,R,R)) -> (B,N,1)`
    logits.  UNet3D + Upsampler are csrc/conv3d.hip; trilinear gather + the 17-layer point MLP are ONE kernel (csrc/sdf_query.hip)."""

    def __init__(self, dim=3, c_dim=128, unet3d=False, unet3d_kwargs=None, upsampler=False, upsampler_kwargs=None,
                 hidden_size=256, n_blocks=5, leaky=False, sample_mode="bilinear", padding=0.1, state_dict=None, device=None,
                 workspace=None):
        _limit(dim == 3 and c_dim == 32 and hidden_size == 32 and n_blocks == 5,
               f"LocalDecoder(dim={dim}, c_dim={c_dim}, hidden_size={hidden_size}, n_blocks={n_blocks}): the fused query kernel is built for 3 / 32 / 32 / 5")
        _limit(not leaky and sample_mode == "bilinear" and abs(padding - 0.1) < 1e-12,
               f"LocalDecoder(leaky={leaky}, sample_mode={sample_mode!r}, padding={padding}): ReLU, trilinear ('bilinear') sampling, padding 0.1")
        uk, pk = unet3d_kwargs or {}, upsampler_kwargs or {}
        d = uk.get("in_channels")
        _limit(bool(unet3d) and uk.get("num_levels") == 3 and d in (64, 128) and uk.get("f_maps") == d and uk.get("out_channels") == d,
               f"LocalDecoder(unet3d={unet3d}, unet3d_kwargs={uk}): a 3-level UNet3D with f_maps = in = out = 64 or 128")
        _limit(bool(upsampler) and pk.get("in_channels") == d and 32 << int(pk.get("upsampler_steps", 0)) == d,
               f"LocalDecoder(upsampler={upsampler}, upsampler_kwargs={pk}): an Upsampler from {d} channels down to 32 (steps = log2(d / 32))")
        self.c_dim, self.n_blocks, self.sample_mode, self.padding = c_dim, n_blocks, sample_mode, padding
        self.d, self.steps = d, int(pk["upsampler_steps"])
        self.dev = _dev_of(device)
        L.lib()
        self.ws = workspace or _Workspace(self.dev)
        self.load_state_dict(_sub_sd(state_dict, "decoder.", 64 >> self.steps))

    def load_state_dict(self, sd):
        dev = self.dev
        self._sd = _np_sd(sd)
        self.unet = {}
        for name in ("encoders.0", "encoders.1", "encoders.2", "decoders.0", "decoders.1"):
            for sc in ("SingleConv1", "SingleConv2"):
                self.unet[f"{name}.{sc}"] = _Conv(sd, f"unet3d.{name}.basic_module.{sc}.", dev, 3, 1, 1)
        self.unet_final = _Conv(sd, "unet3d.final_conv.", dev, 1, 1, 0)
        self.up = []
        for s in range(self.steps):
            self.up.append(_Conv(sd, f"upsampler.blocks.{3 * s + 1}.", dev, 3, 1, 1))
            self.up[-1].pack_subpixel(dev)
            self.up.append(_Conv(sd, f"upsampler.blocks.{3 * s + 2}.", dev, 3, 1, 1))
        from .ops import sdf_pack_weights
        self.sdf_w = torch.from_numpy(sdf_pack_weights(sd, prefix="")).to(dev)

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
            # Conv, ReLU, GroupNorm (updown.py:119-132): the statistics of the GroupNorm are taken in the convolution's epilogue
            x, S = self._conv(x, cv, f"up{i}", sc, sh, up=1 if i % 2 == 0 else 0, relu=True, stats=True)
            sc, sh = self._gn(x, cv.gamma, cv.beta, f"up{i}", partial_S=S)
        if final_affine:
            return self._affine(x, sc, sh, "dec_grid")
        return x, sc, sh

    def query(self, grid_cl, Xtg=None, grid_Q=None, sigmoid=False):
        """Occupancy logits of points Xtg (B,N,3) in [-1,1]^3, or of the makeGrid 'ij' Q^3 lattice, on a decoder feature grid."""
        from . import ops
        if grid_Q is not None:
            axis = torch.from_numpy(np.linspace(-1.0, 1.0, grid_Q).astype(np.float32)).to(self.dev)
            return ops.sdf_query_grid(axis, grid_cl, self.sdf_w, sigmoid=sigmoid)
        return ops.sdf_query(Xtg.to(self.dev, torch.float32), grid_cl, self.sdf_w, sigmoid=sigmoid)

    def forward(self, p, c_grid, **kwargs):
        """dec.py:71-100: p (B,N,3) in [-.5,.5]^3 (= Xtg / 2, vqdif.py:71), c_grid (B,d,R,R,R) -> logits (B,N,1)."""
        c = torch.as_tensor(c_grid).to(self.dev, torch.float32)
        _limit(c.dim() == 5 and c.shape[1] == self.d, f"LocalDecoder feature grid {tuple(c.shape)}: (B,{self.d},R,R,R)")
        grid = self.decoder_grid_cl(c.permute(0, 2, 3, 4, 1).contiguous())
        return self.query(grid, _twice(p, self.dev))

    __call__ = forward


class VQDIF:
    """Inference-side VQDIF (res16: d=128, 2 down/up steps; res32: d=64, 1 step), composed - as the reference composes it,
    vqdif.py:28-32 - of an encoder, a quantizer and a decoder module that share one workspace.  Built either from the latent
    resolution (the shipped configurations) or from three already-instantiated modules (plugin.VQDIFModel: the YAML's
    `encoder_opt` / `quantizer_opt` / `decoder_opt` classes)."""

    G = 64
    # decode_index on a Q^3 lattice: below this Q the decoder grid's last GroupNorm is applied inside the SDF kernel (16 FMAs per point
    # on the interpolated features) instead of as a pass over the 64^3 x 32 grid (67 MB per shape).  Measured round 6: the kernel form
    # costs 2.2 % of the query (124.6 -> 121.9 TFLOP/s), the pass 0.011 ms per shape - equal at 128^3 points per shape, the kernel form
    # 8 x cheaper at 64^3 (BASELINE config 2: 1251 -> 1266 shapes/s), 8 x dearer at 256^3
    AFFINE_IN_QUERY_BELOW_Q = 128
    GROUPS = _GridOps.GROUPS
    EPS = _GridOps.EPS

    def __init__(self, state_dict=None, res=16, device="cuda:0", vocab_size=4096, encoder=None, quantizer=None, decoder=None):
        self.dev = _dev_of(device)
        L.lib()
        if encoder is not None:
            res = encoder.res
        self.res = res
        self.steps = 2 if res == 16 else 1
        self.d = 32 * 2 ** self.steps
        self.K = quantizer.K if quantizer is not None else vocab_size
        self.ws = _Workspace(self.dev)
        sd = state_dict if state_dict is not None else (W.make_state_dict(W.vqdif_spec(res)) if None in (encoder, quantizer, decoder) else None)
        d, steps = self.d, self.steps
        self.encoder = encoder or LocalPoolPointnet(c_dim=32, hidden_dim=32, grid_resolution=64, plane_type="grid", downsampler=True,
                                                    downsampler_kwargs=dict(in_channels=32, downsample_steps=steps), state_dict=sd,
                                                    device=self.dev)
        self.quantizer = quantizer or Quantizer(self.K, d, state_dict=sd, device=self.dev)
        self.decoder = decoder or LocalDecoder(c_dim=32, hidden_size=32, unet3d=True, upsampler=True, sample_mode="bilinear",
                                               unet3d_kwargs=dict(num_levels=3, f_maps=d, in_channels=d, out_channels=d),
                                               upsampler_kwargs=dict(in_channels=d, upsampler_steps=steps), state_dict=sd, device=self.dev)
        _limit(self.encoder.res == res and self.quantizer.d == d and self.decoder.d == d and self.decoder.steps == steps,
               f"VQDIF: encoder latent {self.encoder.res}^3 x {32 << self.encoder.steps}, quantizer n_embd {self.quantizer.d}, decoder "
               f"channels {self.decoder.d} do not describe one autoencoder")
        for m in (self.encoder, self.quantizer, self.decoder):
            if m.dev != self.dev:
                raise L.SfmiError(f"VQDIF: sub-module on {m.dev}, model on {self.dev}")
            m.ws = self.ws                 # one workspace: named buffers are shared, nothing is allocated twice
        if state_dict is not None:                  # handed-in modules take the model's weights; without a state dict they keep their own
            for given, m, prefix in ((encoder, self.encoder, "encoder."), (quantizer, self.quantizer, "quantizer."), (decoder, self.decoder, "decoder.")):
                if given is not None:
                    m.load_state_dict(_sub_sd(state_dict, prefix, res))

    # ------------------------------------------------------------------ weights
    FUSE_DOWN0 = property(lambda self: self.encoder.FUSE_DOWN0, lambda self, v: setattr(self.encoder, "FUSE_DOWN0", v))
    enc_w = property(lambda self: self.encoder.enc_w)
    down = property(lambda self: self.encoder.down)
    last_cell = property(lambda self: self.encoder.last_cell)
    codebook = property(lambda self: self.quantizer.codebook)
    codebook_packed = property(lambda self: self.quantizer.codebook_packed)
    unet = property(lambda self: self.decoder.unet)
    unet_final = property(lambda self: self.decoder.unet_final)
    up = property(lambda self: self.decoder.up)
    sdf_w = property(lambda self: self.decoder.sdf_w)

    def state_dict_np(self):
        """The (numpy, reference-layout) state dict the packed device weights were built from (122 / 110 tensors)."""
        out = {}
        for prefix, m in (("encoder.", self.encoder), ("quantizer.", self.quantizer), ("decoder.", self.decoder)):
            out.update({prefix + k: v for k, v in m._sd.items()})
        return out

    def load_state_dict(self, sd):
        self.encoder.load_state_dict(_sub_sd(sd, "encoder.", self.res))
        self.quantizer.load_state_dict(_sub_sd(sd, "quantizer.", self.res))
        self.decoder.load_state_dict(_sub_sd(sd, "decoder.", self.res))

    # ------------------------------------------------------------------ primitives (delegated to the sub-modules)
    def _buf(self, name, shape, dtype=torch.float32):
        return self.ws.buf(name, shape, dtype)

    def _gn(self, x, gamma, beta, name, **kw):
        return self.decoder._gn(x, gamma, beta, name, **kw)

    def _affine(self, x, sc, sh, name):
        return self.decoder._affine(x, sc, sh, name)

    def _conv(self, *a, **kw):
        return self.decoder._conv(*a, **kw)

    # ------------------------------------------------------------------ encoder (a1-a8)
    def encode_cl(self, cloud):
        """cloud (B,T,3) in [-1,1] -> latent (B,R,R,R,d) channels-last, mask (B,R,R,R) uint8."""
        return self.encoder.encode_cl(cloud)

    def encode(self, Xbd):
        """vqdif.py:35-37 -> (grid_feat (B,d,R,R,R) view, grid_mask (B,R,R,R) bool)."""
        lat, mask = self.encode_cl(Xbd)
        return lat.permute(0, 4, 1, 2, 3).clone(), mask.bool()      # fresh tensors: the *_cl / *_dev paths return workspace views

    # ------------------------------------------------------------------ quantizer (a9, a10)
    def quantize_cl(self, latent):
        return self.quantizer.quantize_cl(latent)

    def get_code_cl(self, ind):
        """quantizer.py:19-30 -> channels-last (B,R,R,R,d)."""
        return self.quantizer.get_code_cl(ind)

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
    def decoder_grid_cl(self, code_cl, final_affine=True):
        """dec.py:75-83: UNet3D + Upsampler -> (B,64,64,64,32) channels-last."""
        return self.decoder.decoder_grid_cl(code_cl, final_affine=final_affine)

    # ------------------------------------------------------------------ a20, a23
    def decode_index(self, code_ind, Xtg=None, grid_Q=None, sigmoid=False, x_range=None):
        """vqdif.py:60-76. Xtg (B,N,3) arbitrary points, or grid_Q=Q for the makeGrid 'ij' Q^3 lattice (x_range = (x0, x1): only its planes
        x0 <= ix < x1 - one rank's slab of dist.sdf_query_sharded)."""
        from . import ops
        if grid_Q is not None and grid_Q < self.AFFINE_IN_QUERY_BELOW_Q:
            # small lattices: the Upsampler's last GroupNorm travels as its (B,32) affine and is applied inside the query kernel
            # (csrc/sdf_query.hip AFF) instead of in a 67 MB-per-shape pass over the 64^3 x 32 grid
            grid, sc, sh = self.decoder_grid_cl(self.get_code_cl(code_ind), final_affine=False)
            axis = torch.from_numpy(np.linspace(-1.0, 1.0, grid_Q).astype(np.float32)).to(self.dev)
            return dict(logits=ops.sdf_query_grid(axis, grid, self.sdf_w, sigmoid=sigmoid, x_range=x_range, affine=(sc, sh)))
        grid = self.decoder_grid_cl(self.get_code_cl(code_ind))
        if grid_Q is not None:
            axis = torch.from_numpy(np.linspace(-1.0, 1.0, grid_Q).astype(np.float32)).to(self.dev)
            return dict(logits=ops.sdf_query_grid(axis, grid, self.sdf_w, sigmoid=sigmoid, x_range=x_range))
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
