"""Training step of the VQDIF autoencoder on the HIP path.

Mirrors `VQDIF.forward/get_loss/training_step/configure_optimizers` (vqdif.py:78-137), `VQLoss` (:151-167) and the
training branch of `Quantizer.forward` (quantizer.py:31-89: nearest code from the PRE-update codebook, EMA update of
N / z_avg / embedding.weight with gamma .99, straight-through gradient, commitment term) for the shipped res16 / res32
configurations.  Everything runs through libsfmi (csrc/train_vqdif.hip + the inference kernels); torch supplies device
memory, a few weight re-layouts (transpose / tap flip) and O(B*C) GroupNorm coefficient algebra.  No autograd: the
forward records a tape of closures and `backward` replays it.  Data parallel: gradients live in one flat buffer that is
all-reduced (mean) before Adam, and the EMA statistics (code counts and sums) are all-reduced (sum) so every rank keeps
the same codebook (the reference's DDP does not sync them - its ranks' codebooks drift; documented deviation).

Layouts: point features (rows, C); grids channels-last (B, D, H, W, C); conv weights are kept as [tap][Cout][Cin]
masters (`state_dict()` converts back to the reference's (Cout, Cin, k, k, k)).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L

G = 64   # encoder / decoder feature grid


def _ck(rc, what):
    L.check(rc, what)


class Tape:
    def __init__(self):
        self.ops, self.grads = [], {}

    def add(self, out, inputs, fn):
        self.ops.append((out, inputs, fn))

    def backward(self, out, dout, add):
        self.grads = {id(out): dout}
        for o, inputs, fn in reversed(self.ops):
            g = self.grads.pop(id(o), None)
            if g is None:
                continue
            for t, d in zip(inputs, fn(g)):
                if t is None or d is None:
                    continue
                k = id(t)
                self.grads[k] = add(self.grads[k], d) if k in self.grads else d
        self.ops = []


class VQDIFTrainer:
    def __init__(self, state_dict, res=16, device="cuda:0", lr=1e-4, beta=0.001, gamma=0.99, dist=None, n_groups=8):
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.SfmiError("VQDIFTrainer needs a HIP device (no CPU fallback)")
        self.lib = L.lib()
        self.res, self.lr, self.beta, self.gamma, self.dist, self.ng = res, lr, beta, gamma, dist, n_groups
        self.steps = 2 if res == 16 else 1
        self.d = 32 << self.steps
        self.step_count = 0
        # ---- parameters: one flat buffer (params / grads / Adam moments), views per tensor -------------------------
        self.shapes, self.kind = {}, {}
        host = {}
        for k, v in state_dict.items():
            v = np.asarray(v, np.float32)
            if k.startswith("quantizer."):
                continue
            if k.endswith("conv.weight") or k.endswith("final_conv.weight"):
                co, ci, ks = v.shape[0], v.shape[1], v.shape[2]
                v = np.ascontiguousarray(v.reshape(co, ci, ks ** 3).transpose(2, 0, 1))     # [tap][Cout][Cin]
                self.kind[k] = ("conv", ks)
            host[k] = v
        self.names = sorted(host)
        al = lambda k: (k + 3) // 4 * 4                          # every tensor starts 16-byte aligned (float4 loads in the GEMMs)
        n = sum(al(v.size) for v in host.values())
        self.flat_p = torch.zeros(n, device=self.dev)
        self.flat_g = torch.zeros(n, device=self.dev)
        self.flat_m = torch.zeros(n, device=self.dev)
        self.flat_v = torch.zeros(n, device=self.dev)
        self.p, self.g = {}, {}
        o = 0
        for k in self.names:
            v = host[k]
            self.flat_p[o:o + v.size] = torch.from_numpy(v.reshape(-1)).to(self.dev)
            self.p[k] = self.flat_p[o:o + v.size].view(v.shape)
            self.g[k] = self.flat_g[o:o + v.size].view(v.shape)
            o += al(v.size)
        self.K = state_dict["quantizer.embedding.weight"].shape[0]
        self.emb = torch.from_numpy(np.asarray(state_dict["quantizer.embedding.weight"], np.float32)).to(self.dev).contiguous()
        self.N = torch.from_numpy(np.asarray(state_dict["quantizer.N"], np.float32)).to(self.dev).contiguous()
        self.z_avg = torch.from_numpy(np.asarray(state_dict["quantizer.z_avg"], np.float32)).to(self.dev).contiguous()

    # ------------------------------------------------------------------------------------------------ helpers
    def _f(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.float32)

    def _add(self, a, b):
        out = self._f(*a.shape)
        _ck(self.lib.sfmi_add_f32(L.ptr(a), L.ptr(b), L.ptr(out), a.numel(), L.stream_ptr()), "add")
        return out

    def _gemm(self, x, W, bias, resid, M, N, K, act=0):
        y = self._f(M, N)
        _ck(self.lib.sfmi_gemm_f32(L.ptr(x), L.ptr(W), L.ptr(bias), L.ptr(resid), L.ptr(y), M, N, K, act, 0, 0, L.stream_ptr()), "gemm")
        return y

    def _relu(self, x):
        y = self._f(*x.shape)       # relu(x) == relu_bwd(dy = x, y = x)
        _ck(self.lib.sfmi_relu_bwd_f32(L.ptr(x), L.ptr(x), L.ptr(y), x.numel(), L.stream_ptr()), "relu")
        self._tap_relu(y)
        return y

    def _tap_relu(self, y):
        """Debug tap (tests/test_train_vqdif_gpu.py): with `self.relu_tap = []` set, the activation pattern (y > 0) of every
        ReLU of the forward is recorded in call order, for a frozen-mask gradient comparison against the oracle."""
        if getattr(self, "relu_tap", None) is not None:
            self.relu_tap.append((y > 0).cpu())

    def _relu_bwd(self, dy, y):
        dx = self._f(*dy.shape)
        _ck(self.lib.sfmi_relu_bwd_f32(L.ptr(dy), L.ptr(y), L.ptr(dx), dy.numel(), L.stream_ptr()), "relu_bwd")
        return dx

    def _colsum_into(self, x, M, N, gname):
        if M >= 256:
            ws = self._f(self.lib.sfmi_colsum_slices(M, N) * N)
            _ck(self.lib.sfmi_colsum_ws_f32(L.ptr(x), L.ptr(self.g[gname]), M, N, N, 0, L.ptr(ws), L.stream_ptr()), "colsum_ws")
        else:
            _ck(self.lib.sfmi_colsum_f32(L.ptr(x), L.ptr(self.g[gname]), M, N, N, 0, L.stream_ptr()), "colsum")

    def _wgrad_into(self, dy, x, gname, B, Di, Hi, Wi, Cin, Cout, KS, stride, pad, ldy=None, ldx=None, rows=None):
        """dW[tap][co][ci] of a channels-last conv (KS == 1: a Linear over `Wi` rows) -> self.g[gname] (first Cin cols)."""
        Do = (Di + 2 * pad - KS) // stride + 1
        rows = B * Do * ((Hi + 2 * pad - KS) // stride + 1) * ((Wi + 2 * pad - KS) // stride + 1)
        tiles = -(-Cout // 64) * -(-Cin // 64) * KS ** 3
        nsplit = max(1, min(512, -(-1024 // tiles), rows // 256))
        part = self._f(nsplit, KS ** 3 * Cout * Cin)
        _ck(self.lib.sfmi_conv3d_wgrad_f32(L.ptr(dy), L.ptr(x), L.ptr(part), B, Di, Hi, Wi, Cin, Cout, KS, stride, pad,
                                           ldy or Cout, ldx or Cin, nsplit, L.stream_ptr()), "wgrad")
        g = self.g[gname]
        assert g.numel() == KS ** 3 * Cout * Cin, (gname, g.shape, Cout, Cin)
        _ck(self.lib.sfmi_colsum_f32(L.ptr(part), L.ptr(g), nsplit, g.numel(), g.numel(), 0, L.stream_ptr()), "colsum")

    # ------------------------------------------------------------------------------------------------ taped ops
    def linear(self, x, wname, bname=None, act=0, resid=None, need_dx=True):
        """y = act(x W^T + b) (+ resid).  x (M,K) with K % 16 == 0 (pad narrower inputs before), W (N,K), N % 32 == 0."""
        W, b = self.p[wname], self.p[bname] if bname else None
        M, K = x.shape
        N = W.shape[0]
        Wk = W
        if W.shape[1] != K:                      # input was zero-padded (fc_pos / fc_p: K = 3 -> 16)
            Wk = torch.zeros(N, K, device=self.dev)
            Wk[:, :W.shape[1]] = W
        y = self._gemm(x, Wk, b, resid, M, N, K, act)
        if act == 1:
            self._tap_relu(y)

        def bwd(dy):
            if act == 1:
                assert resid is None
                dy = self._relu_bwd(dy, y)
            Kw = W.shape[1]
            self._wgrad_into(dy, x, wname, 1, 1, 1, M, Kw, N, 1, 1, 0, ldy=N, ldx=K)
            if bname:
                self._colsum_into(dy, M, N, bname)
            dx = self._gemm(dy, W.t().contiguous(), None, None, M, K, N) if need_dx else None
            return [dx, dy if resid is not None else None]
        self.tape.add(y, [x, resid], bwd)
        return y

    def resblock(self, x, prefix, need_dx=True):
        """ResnetBlockFC (layers.py:6-48): x_s + fc_1(relu(fc_0(relu(x)))), x_s = shortcut(x) iff in != out."""
        a0 = self._relu(x)
        self.tape.add(a0, [x], lambda d: [self._relu_bwd(d, a0)] if need_dx else [None])
        a1 = self.linear(a0, prefix + "fc_0.weight", prefix + "fc_0.bias", act=1, need_dx=need_dx)
        sk = prefix + "shortcut.weight"
        xs = self.linear(x, sk, None, need_dx=need_dx) if sk in self.p else x
        return self.linear(a1, prefix + "fc_1.weight", prefix + "fc_1.bias", resid=xs)

    def conv(self, x, wname, B, Di, Cin, Cout, KS, stride, pad, relu, bias=None, need_dx=True):
        """Conv3d on a cubic channels-last grid (+ fused ReLU); x (B,Di,Di,Di,Cin)."""
        w = self.p[wname]
        Do = (Di + 2 * pad - KS) // stride + 1
        y = self._f(B, Do, Do, Do, Cout)
        _ck(self.lib.sfmi_conv3d_cl_f32(L.ptr(x), L.ptr(w), None, None, L.ptr(self.p[bias]) if bias else None, L.ptr(y), B, Di, Di, Di,
                                        Cin, Cout, KS, stride, pad, 0, int(relu), L.stream_ptr()), "conv")
        if relu:
            self._tap_relu(y)

        def bwd(dy):
            if relu:
                dy = self._relu_bwd(dy, y)
            self._wgrad_into(dy, x, wname, B, Di, Di, Di, Cin, Cout, KS, stride, pad)
            if bias:
                self._colsum_into(dy, B * Do ** 3, Cout, bias)
            if not need_dx:
                return [None]
            dx = self._f(B, Di, Di, Di, Cin)
            if KS == 3:      # stride 1, pad 1: correlation with the tap-flipped, channel-transposed kernel
                wd = w.flip(0).transpose(1, 2).contiguous()
                _ck(self.lib.sfmi_conv3d_cl_f32(L.ptr(dy), L.ptr(wd), None, None, None, L.ptr(dx), B, Do, Do, Do, Cout, Cin, 3, 1, 1, 0, 0,
                                                L.stream_ptr()), "conv dgrad")
            elif KS == 1:
                wd = w[0].t().contiguous()
                _ck(self.lib.sfmi_gemm_f32(L.ptr(dy), L.ptr(wd), None, None, L.ptr(dx), B * Do ** 3, Cin, Cout, 0, 0, 0, L.stream_ptr()), "1x1 dgrad")
            else:            # k2 s2: every input voxel belongs to exactly one (output voxel, tap)
                dxv = dx.view(B, Do, 2, Do, 2, Do, 2, Cin)
                for t in range(8):
                    wd = w[t].t().contiguous()                      # (Cin, Cout)
                    tmp = self._gemm(dy, wd, None, None, B * Do ** 3, Cin, Cout)
                    dxv[:, :, t >> 2, :, (t >> 1) & 1, :, t & 1, :] = tmp.view(B, Do, Do, Do, Cin)
            return [dx]
        self.tape.add(y, [x], bwd)
        return y

    def groupnorm(self, x, gname, bname, B, V, C):
        """nn.GroupNorm(8, C) on (B,V,C); backward is affine in (dy, x) per (sample, channel)."""
        lib, S = self.lib, self.lib.sfmi_gn_splits(V)
        gamma, beta = self.p[gname], self.p[bname]
        scale, shift = self._f(B, C), self._f(B, C)
        part = torch.empty(B, S, C, 2, device=self.dev, dtype=torch.float64)
        _ck(lib.sfmi_groupnorm_coeffs_f32(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(scale), L.ptr(shift), L.ptr(part), B, V, C, self.ng,
                                          1e-5, L.stream_ptr()), "gn coeffs")
        y = self._f(*x.shape)
        _ck(lib.sfmi_affine_cl_f32(L.ptr(x), L.ptr(scale), L.ptr(shift), L.ptr(y), B, V, C, L.stream_ptr()), "affine")

        def bwd(dy):
            cg = C // self.ng
            n = float(V * cg)
            sx = part.sum(1)                                                   # (B,C,2): sum x, sum x^2
            mu = sx[..., 0].view(B, self.ng, cg).sum(-1) / n                   # (B,G)
            var = sx[..., 1].view(B, self.ng, cg).sum(-1) / n - mu * mu
            rstd = 1.0 / torch.sqrt(var.clamp_min(0) + 1e-5)
            pd = torch.empty(B, S, C, 2, device=self.dev, dtype=torch.float64)
            _ck(lib.sfmi_chan_dot_stats_f32(L.ptr(dy), L.ptr(x), L.ptr(pd), B, V, C, S, L.stream_ptr()), "chan_dot_stats")
            sd = pd.sum(1)
            s1, s2 = sd[..., 0], sd[..., 1]                                    # (B,C): sum dy, sum dy*x
            gm = gamma.double()
            mu_c, rstd_c = mu.repeat_interleave(cg, 1), rstd.repeat_interleave(cg, 1)
            xh = (s2 - mu_c * s1) * rstd_c                                     # sum dy * xhat
            m1 = (gm * s1).view(B, self.ng, cg).sum(-1) / n
            m2 = (gm * xh).view(B, self.ng, cg).sum(-1) / n
            A = (gm[None] * rstd_c).float().contiguous()
            Bc = (-(rstd * rstd * m2)).repeat_interleave(cg, 1).float().contiguous()
            Cc = (rstd * rstd * m2 * mu - rstd * m1).repeat_interleave(cg, 1).float().contiguous()
            self.g[gname].copy_(xh.sum(0).float())
            self.g[bname].copy_(s1.sum(0).float())
            dx = self._f(*x.shape)
            _ck(lib.sfmi_affine2_cl_f32(L.ptr(dy), L.ptr(x), L.ptr(A), L.ptr(Bc), L.ptr(Cc), L.ptr(dx), B, V, C, L.stream_ptr()), "affine2")
            return [dx]
        self.tape.add(y, [x], bwd)
        return y

    def maxpool(self, x, B, Do, C):
        y = self._f(B, Do, Do, Do, C)
        _ck(self.lib.sfmi_maxpool2_cl_f32(L.ptr(x), L.ptr(y), B, Do, Do, Do, C, L.stream_ptr()), "maxpool")
        if getattr(self, "pool_tap", None) is not None:   # debug tap: the selected window element (first maximum, 4 dz + 2 dy + dx)
            win = x.view(B, Do, 2, Do, 2, Do, 2, C).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(B, Do, Do, Do, C, 8)
            first = (win == y.unsqueeze(-1)).int() * torch.arange(8, 0, -1, device=self.dev, dtype=torch.int32)
            self.pool_tap.append(first.argmax(-1).cpu())

        def bwd(dy):
            dx = self._f(*x.shape)
            _ck(self.lib.sfmi_maxpool2_bwd_cl_f32(L.ptr(x), L.ptr(y), L.ptr(dy), L.ptr(dx), B, Do, Do, Do, C, L.stream_ptr()), "maxpool bwd")
            return [dx]
        self.tape.add(y, [x], bwd)
        return y

    def upcat(self, skip, low, B, D, Cs, Cu):
        """cat(skip, nearest_x2(low)) on channels (unet3d.py:268-293); Cs == 0: plain nearest x2 (updown.py:121)."""
        y = self._f(B, D, D, D, Cs + Cu)
        if Cs:
            _ck(self.lib.sfmi_upcat_cl_f32(L.ptr(skip), L.ptr(low), L.ptr(y), B, D, D, D, Cs, Cu, L.stream_ptr()), "upcat")
        else:
            _ck(self.lib.sfmi_upsample2_cl_f32(L.ptr(low), L.ptr(y), B, D // 2, D // 2, D // 2, Cu, L.stream_ptr()), "upsample2")

        def bwd(dy):
            dlow = self._f(B, D // 2, D // 2, D // 2, Cu)
            _ck(self.lib.sfmi_sumpool2_cl_f32(L.ptr(dy), L.ptr(dlow), B, D // 2, D // 2, D // 2, Cs + Cu, Cs, Cu, L.stream_ptr()), "sumpool2")
            return [dy[..., :Cs].contiguous() if Cs else None, dlow]
        self.tape.add(y, [skip, low], bwd)
        return y

    # ------------------------------------------------------------------------------------------------ forward pieces
    def _encoder(self, Xbd):
        lib, B, T = self.lib, Xbd.shape[0], Xbd.shape[1]
        M, ncell = B * T, G ** 3
        cell = torch.empty(B, T, device=self.dev, dtype=torch.int32)
        ph = self._f(M, 3)
        _ck(lib.sfmi_cells_f32(L.ptr(Xbd), L.ptr(cell), L.ptr(ph), B, T, G, L.stream_ptr()), "cells")
        p16 = torch.zeros(M, 16, device=self.dev)
        p16[:, :3] = ph
        net = self.linear(p16, "encoder.fc_pos.weight", "encoder.fc_pos.bias", need_dx=False)            # (M,64)
        net = self.resblock(net, "encoder.blocks.0.")
        for i in range(1, 5):
            keys = torch.empty(B, ncell, 32, device=self.dev, dtype=torch.int32)
            keys.view(torch.uint8).fill_(0x80)
            cat = self._f(M, 64)
            cat[:, :32] = net
            _ck(lib.sfmi_cell_max_f32(L.ptr(net), L.ptr(cell), L.ptr(keys), L.ptr(cat), B, T, ncell, 32, 64, 32, L.stream_ptr()), "cell_max")

            def bwd(dcat, net=net, keys=keys):
                # left half: identity; right half: pooled -> scatter to cells, route to the arg-max points
                acc = torch.zeros(B, ncell, 32, device=self.dev, dtype=torch.int64)
                _ck(lib.sfmi_cell_scatter_add_f32(L.ptr(dcat), L.ptr(cell), L.ptr(acc), None, B, T, ncell, 32, 64, 32, L.stream_ptr()), "cell_scatter")
                dnet = dcat[:, :32].contiguous()
                _ck(lib.sfmi_cell_max_bwd_f32(L.ptr(net), L.ptr(keys), L.ptr(acc), L.ptr(cell), L.ptr(dnet), B, T, ncell, 32, 32, 1,
                                              L.stream_ptr()), "cell_max_bwd")
                return [dnet]
            self.tape.add(cat, [net], bwd)
            net = self.resblock(cat, f"encoder.blocks.{i}.")
        c = self.linear(net, "encoder.fc_c.weight", "encoder.fc_c.bias")                                   # (M,32)
        acc = torch.zeros(B, ncell, 32, device=self.dev, dtype=torch.int64)
        cnt = torch.zeros(B, ncell, device=self.dev, dtype=torch.int32)
        _ck(lib.sfmi_cell_scatter_add_f32(L.ptr(c), L.ptr(cell), L.ptr(acc), L.ptr(cnt), B, T, ncell, 32, 32, 0, L.stream_ptr()), "cell_scatter")
        grid = self._f(B, G, G, G, 32)
        _ck(lib.sfmi_cell_mean_f32(L.ptr(acc), L.ptr(cnt), L.ptr(grid), B, ncell, 32, L.stream_ptr()), "cell_mean")

        def bwd_mean(dgrid):
            dc = self._f(M, 32)
            _ck(lib.sfmi_cell_mean_bwd_f32(L.ptr(dgrid), L.ptr(cnt), L.ptr(cell), L.ptr(dc), B, T, ncell, 32, L.stream_ptr()), "cell_mean_bwd")
            return [dc]
        self.tape.add(grid, [c], bwd_mean)
        # Downsampler (updown.py:101-118): steps x [conv k2 s2 -> ReLU -> GN ; conv 1x1 -> ReLU -> GN], channels double
        x, D, C = grid, G, 32
        for s in range(self.steps):
            pre = f"encoder.downsampler.blocks.{2 * s}."
            x = self.conv(x, pre + "conv.weight", B, D, C, 2 * C, 2, 2, 0, True)
            D, C = D // 2, 2 * C
            x = self.groupnorm(x, pre + "groupnorm.weight", pre + "groupnorm.bias", B, D ** 3, C)
            pre = f"encoder.downsampler.blocks.{2 * s + 1}."
            x = self.conv(x, pre + "conv.weight", B, D, C, C, 1, 1, 0, True)
            x = self.groupnorm(x, pre + "groupnorm.weight", pre + "groupnorm.bias", B, D ** 3, C)
        return x                                                                                             # (B,R,R,R,d)

    def _quantize(self, latent):
        """Nearest code of the current codebook, straight-through output, commitment loss (quantizer.py:31-89)."""
        lib, d = self.lib, self.d
        rows = latent.numel() // d
        W = self.emb
        pk = torch.cat([W.view(self.K // 32, 32, d // 8, 2, 4).permute(0, 2, 3, 1, 4).reshape(-1), (W * W).sum(1)]).contiguous()
        idx = torch.empty(rows, device=self.dev, dtype=torch.int32)
        _ck(lib.sfmi_vq_argmin_f32(L.ptr(latent), L.ptr(pk), L.ptr(idx), None, rows, self.K, d, L.stream_ptr()), "vq_argmin")
        q = self._f(*latent.shape)
        _ck(lib.sfmi_vq_gather_f32(L.ptr(W), L.ptr(idx), L.ptr(q), rows, d, L.stream_ptr()), "vq_gather")
        diffv = self._f(latent.numel())
        _ck(lib.sfmi_lincomb_f32(1.0, L.ptr(latent), -1.0, L.ptr(q), L.ptr(diffv), latent.numel(), L.stream_ptr()), "x - q")
        diff = (diffv.double() ** 2).mean()                               # scalar for the loss value (logging)
        cscale = 2.0 * self.beta / latent.numel()

        def bwd(dq):   # d loss / d latent = straight-through + beta * d mean((x - q)^2) / dx
            dl = self._f(*latent.shape)
            _ck(lib.sfmi_lincomb_f32(1.0, L.ptr(dq), cscale, L.ptr(diffv), L.ptr(dl), latent.numel(), L.stream_ptr()), "st + commit")
            return [dl]
        self.tape.add(q, [latent], bwd)
        return q, idx, diff

    def _single_gcr(self, x, prefix, B, D, Cin, Cout):
        x = self.groupnorm(x, prefix + "groupnorm.weight", prefix + "groupnorm.bias", B, D ** 3, Cin)
        return self.conv(x, prefix + "conv.weight", B, D, Cin, Cout, 3, 1, 1, True)

    def _double(self, x, prefix, B, D, Cin):
        c1 = self.p[prefix + "SingleConv1.conv.weight"].shape[1]
        c2 = self.p[prefix + "SingleConv2.conv.weight"].shape[1]
        return self._single_gcr(self._single_gcr(x, prefix + "SingleConv1.", B, D, Cin, c1), prefix + "SingleConv2.", B, D, c1, c2), c2

    def _decoder_grid(self, q, B):
        R, d, P = self.res, self.d, "decoder.unet3d."
        e0, c0 = self._double(q, P + "encoders.0.basic_module.", B, R, d)
        e1, c1 = self._double(self.maxpool(e0, B, R // 2, c0), P + "encoders.1.basic_module.", B, R // 2, c0)
        e2, c2 = self._double(self.maxpool(e1, B, R // 4, c1), P + "encoders.2.basic_module.", B, R // 4, c1)
        y, cy = self._double(self.upcat(e1, e2, B, R // 2, c1, c2), P + "decoders.0.basic_module.", B, R // 2, c1 + c2)
        y, cy = self._double(self.upcat(e0, y, B, R, c0, cy), P + "decoders.1.basic_module.", B, R, c0 + cy)
        y = self.conv(y, P + "final_conv.weight", B, R, cy, d, 1, 1, 0, False, bias=P + "final_conv.bias")
        D, C = R, d
        for s in range(self.steps):       # Upsampler (updown.py:119-132)
            y = self.upcat(None, y, B, 2 * D, 0, C)
            D = 2 * D
            for j in (1, 2):
                pre = f"decoder.upsampler.blocks.{3 * s + j}."
                co = C // 2 if j == 1 else C
                y = self.conv(y, pre + "conv.weight", B, D, C, co, 3, 1, 1, True)
                C = co
                y = self.groupnorm(y, pre + "groupnorm.weight", pre + "groupnorm.bias", B, D ** 3, C)
        return y                                                                                             # (B,64,64,64,32)

    def _sdf_head(self, grid, Xtg):
        lib, B, N = self.lib, Xtg.shape[0], Xtg.shape[1]
        M = B * N
        c = self._f(M, 32)
        _ck(lib.sfmi_trilinear_cl_f32(L.ptr(Xtg), L.ptr(grid), L.ptr(c), B, N, G, 32, L.stream_ptr()), "trilinear")

        def bwd(dc):
            acc = torch.zeros(B, G ** 3, 32, device=self.dev, dtype=torch.int64)
            _ck(lib.sfmi_trilinear_bwd_cl_f32(L.ptr(Xtg), L.ptr(dc), L.ptr(acc), B, N, G, 32, L.stream_ptr()), "trilinear bwd")
            dgrid = self._f(B, G, G, G, 32)
            _ck(lib.sfmi_fixed_to_float_f32(L.ptr(acc), L.ptr(dgrid), acc.numel(), 0, L.stream_ptr()), "fixed_to_float")
            return [dgrid]
        self.tape.add(c, [grid], bwd)
        p16 = torch.zeros(M, 16, device=self.dev)
        p16[:, :3] = Xtg.reshape(M, 3) * 0.5                                                                 # dec.py:88 p = Xtg / 2
        net = self.linear(p16, "decoder.fc_p.weight", "decoder.fc_p.bias", need_dx=False)
        for i in range(5):
            net = self.linear(c, f"decoder.fc_c.{i}.weight", f"decoder.fc_c.{i}.bias", resid=net)
            net = self.resblock(net, f"decoder.blocks.{i}.")
        a = self._relu(net)
        self.tape.add(a, [net], lambda d: [self._relu_bwd(d, a)])
        # fc_out (1,32): padded to 32 output columns for the GEMM tile; only column 0 is real
        Wp = torch.zeros(32, 32, device=self.dev)
        Wp[0] = self.p["decoder.fc_out.weight"][0]
        bp = torch.zeros(32, device=self.dev)
        bp[0] = self.p["decoder.fc_out.bias"][0]
        y = self._gemm(a, Wp, bp, None, M, 32, 32)
        logits = y[:, 0].contiguous()

        def bwd_out(dlog):
            dY = torch.zeros(M, 32, device=self.dev)
            dY[:, 0] = dlog
            gw = self._f(1, 32 * 32)
            _ck(lib.sfmi_conv3d_wgrad_f32(L.ptr(dY), L.ptr(a), L.ptr(gw), 1, 1, 1, M, 32, 32, 1, 1, 0, 32, 32, 1, L.stream_ptr()), "wgrad fc_out")
            self.g["decoder.fc_out.weight"].copy_(gw.view(32, 32)[0:1])
            self.g["decoder.fc_out.bias"].copy_(dlog.sum().view(1))
            return [self._gemm(dY, Wp.t().contiguous(), None, None, M, 32, 32)]
        self.tape.add(logits, [a], bwd_out)
        return logits

    # ------------------------------------------------------------------------------------------------ step
    @torch.no_grad()
    def loss_and_grad(self, Xbd, Xtg, Ytg):
        """-> dict(loss, recon_loss, diff_loss); gradients of every parameter in self.g (flat buffer self.flat_g)."""
        lib = self.lib
        self.tape = Tape()
        self.flat_g.zero_()
        Xbd = torch.as_tensor(Xbd).to(self.dev, torch.float32).contiguous()
        Xtg = torch.as_tensor(Xtg).to(self.dev, torch.float32).contiguous()
        Ytg = torch.as_tensor(Ytg).to(self.dev, torch.float32).reshape(-1).contiguous()
        B = Xbd.shape[0]
        latent = self._encoder(Xbd)
        q, idx, diff = self._quantize(latent)
        self._last = (latent, idx)
        logits = self._sdf_head(self._decoder_grid(q, B), Xtg)
        n = logits.numel()
        rows, dlog = self._f(n), self._f(n)
        _ck(lib.sfmi_bce_logits_f32(L.ptr(logits), L.ptr(Ytg), L.ptr(rows), L.ptr(dlog), n, 1.0 / n, L.stream_ptr()), "bce")
        recon = rows.double().mean()
        self.tape.backward(logits, dlog, self._add)
        return dict(loss=(recon + self.beta * diff).float(), recon_loss=recon.float(), diff_loss=diff.float(), logits=logits)

    @torch.no_grad()
    def ema_update(self):
        """quantizer.py:68-86 on the statistics of the last forward (summed over ranks under data parallelism)."""
        lib = self.lib
        latent, idx = self._last
        rows = latent.numel() // self.d
        sums = torch.zeros(self.K, self.d, device=self.dev, dtype=torch.int64)
        cnts = torch.zeros(self.K, device=self.dev, dtype=torch.int32)
        _ck(lib.sfmi_vq_stats_f32(L.ptr(latent), L.ptr(idx), L.ptr(sums), L.ptr(cnts), rows, self.d, L.stream_ptr()), "vq_stats")
        sf = self._f(self.K, self.d)
        _ck(lib.sfmi_fixed_to_float_f32(L.ptr(sums), L.ptr(sf), sums.numel(), 0, L.stream_ptr()), "fixed_to_float")
        cf = cnts.float()
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            self.dist.all_reduce(sf)
            self.dist.all_reduce(cf)
        _ck(lib.sfmi_vq_ema_update_f32(L.ptr(self.N), L.ptr(self.z_avg), L.ptr(self.emb), L.ptr(cf), L.ptr(sf), self.K, self.d, self.gamma, 1e-7,
                                       L.stream_ptr()), "vq_ema")

    @torch.no_grad()
    def optimizer_step(self):
        """torch.optim.Adam(lr) over all parameters (vqdif.py:121-126) == AdamW kernel with weight decay 0, betas (.9,.999)."""
        from .dist import allreduce_mean_
        allreduce_mean_(self.flat_g, self.dist)
        self.step_count += 1
        _ck(self.lib.sfmi_adamw_f32(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.flat_m), L.ptr(self.flat_v), self.flat_p.numel(),
                                    self.lr, 0.9, 0.999, 1e-8, 0.0, self.step_count, L.stream_ptr()), "adam")

    def training_step(self, batch):
        """VQDIF.training_step (vqdif.py:100-105) + optimizer + EMA: batch = {Xbd, Xtg, Ytg} -> losses dict."""
        out = self.loss_and_grad(batch["Xbd"], batch["Xtg"], batch["Ytg"])
        self.ema_update()
        self.optimizer_step()
        return out

    def optimizer_state(self):
        return dict(step=self.step_count, exp_avg=self.flat_m.detach().cpu(), exp_avg_sq=self.flat_v.detach().cpu(), names=list(self.names))

    def load_optimizer_state(self, st):
        assert st["names"] == list(self.names), "optimizer state does not match this parameter table"
        self.step_count = int(st["step"])
        self.flat_m.copy_(st["exp_avg"].to(self.dev))
        self.flat_v.copy_(st["exp_avg_sq"].to(self.dev))

    def state_dict(self):
        """Reference layouts / key names (usable by VQDIF.load_state_dict and torch.save as a Lightning `state_dict`)."""
        sd = {}
        for k in self.names:
            v = self.p[k].detach().cpu().numpy()
            if k in self.kind:
                ks = self.kind[k][1]
                v = np.ascontiguousarray(v.transpose(1, 2, 0)).reshape(v.shape[1], v.shape[2], ks, ks, ks)
            sd[k] = v.copy()
        sd["quantizer.embedding.weight"], sd["quantizer.N"], sd["quantizer.z_avg"] = (t.cpu().numpy() for t in (self.emb, self.N, self.z_avg))
        return sd
