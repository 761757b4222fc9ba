"""CPU oracle for the VQDIF half of the hot path (encode -> quantize -> decode).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the *checker*; the product path
(shapeformer_amd/) never imports it.

It is a functional restatement (plain torch-CPU fp32 ops on explicit tensors,
weights passed as a {key: tensor} state dict with the reference's key names) of
the reference's VQDIF modules; every function cites the reference lines it
follows.  Pinned against the imported reference itself by
oracle/make_golden.py (max |diff| == 0 for integer outputs, fp32 roundoff for
float outputs) and against the committed vectors in tests/golden/.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

PAD = 0.1
# Frozen-activation hook for gradient parity tests: two fp32 implementations of a ReLU network take different branches on units
# whose pre-activation is within rounding error of zero, which changes those units' whole gradient contribution.  When a list
# of boolean masks is installed here (tests only; oracle.vqdif_train_oracle.loss_and_grads(relu_masks=...)), every ReLU of the
# forward - consumed in call order - becomes x * mask, i.e. both implementations differentiate the SAME piecewise-linear
# function and their gradients must then agree to rounding error.  Masks of conv outputs arrive channels-last.
RELU_MASKS = None


def _relu(x):
    if RELU_MASKS is None:
        return F.relu(x)
    m = RELU_MASKS.pop(0)
    if x.dim() == 5:
        m = m.permute(0, 4, 1, 2, 3)
    assert m.numel() == x.numel(), (tuple(m.shape), tuple(x.shape))
    return x * m.reshape(x.shape).to(x.dtype)


POOL_INDEX = None   # same idea for the UNet's 2^3 max-pools: which of the 8 window elements (4 dz + 2 dy + dx) is taken, channels-last


def _max_pool2(x):
    """F.max_pool3d(x, 2), or - with POOL_INDEX installed - the window element another implementation's forward selected (a
    post-ReLU window whose only live unit is within rounding error of zero has its arg-max on that unit in one implementation
    and on a dead neighbour in the other: the same kind of branch flip as a ReLU's)."""
    if POOL_INDEX is None:
        return F.max_pool3d(x, 2)
    idx = POOL_INDEX.pop(0).permute(0, 4, 1, 2, 3).long()                      # (B,C,Do,Ho,Wo)
    B, C, D, H, W = x.shape
    win = x.view(B, C, D // 2, 2, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(B, C, D // 2, H // 2, W // 2, 8)
    return win.gather(-1, idx.unsqueeze(-1)).squeeze(-1)


G = 64  # encoder / decoder feature grid (configs/vqdif/*.yaml grid_resolution)


def to_torch_sd(sd):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}


# ---- a2 / a3 -------------------------------------------------------------
def normalize_3d(p, padding=PAD):
    """vqdif/common.py:260-276: p/(1+pad+1e-3)+0.5, clamp >=1 -> 1-1e-3, <0 -> 0."""
    u = p / (1 + padding + 10e-4) + 0.5
    u = torch.where(u >= 1, torch.full_like(u, 1 - 10e-4), u)
    u = torch.where(u < 0, torch.zeros_like(u), u)
    return u


def cell_index(u, reso=G):
    """vqdif/common.py:300-321 ('original' order): x + R*(y + R*z), trunc toward 0."""
    i = (u * reso).long()
    return i[..., 0] + reso * (i[..., 1] + reso * i[..., 2])


# ---- a4 --------------------------------------------------------------------
def resblock(sd, prefix, x):
    """vqdif/layers.py:39-48: x_s + fc_1(relu(fc_0(relu(x))))."""
    h = F.linear(_relu(x), sd[prefix + "fc_0.weight"], sd[prefix + "fc_0.bias"])
    dx = F.linear(_relu(h), sd[prefix + "fc_1.weight"], sd[prefix + "fc_1.bias"])
    sk = prefix + "shortcut.weight"
    xs = F.linear(x, sd[sk]) if sk in sd else x
    return xs + dx


# ---- a6 --------------------------------------------------------------------
def local_max_pool(net, cell, ncell=G ** 3):
    """enc.py:95-112 (+torch_scatter.scatter_max): per-cell channel max gathered back."""
    B, T, C = net.shape
    idx = cell[:, :, None].expand(B, T, C)
    grid = torch.full((B, ncell, C), float("-inf"), dtype=net.dtype)
    grid = grid.scatter_reduce(1, idx, net, reduce="amax", include_self=True)
    return grid.gather(1, idx)


def conv_relu_gn(sd, prefix, x, stride, padding):
    """updown.py:79-99 order 'crg': conv(no bias) -> ReLU -> GroupNorm(8)."""
    x = F.conv3d(x, sd[prefix + "conv.weight"], None, stride=stride, padding=padding)
    x = _relu(x)
    return F.group_norm(x, 8, sd[prefix + "groupnorm.weight"], sd[prefix + "groupnorm.bias"], 1e-5)


def downsampler(sd, x, prefix="encoder.downsampler.blocks."):
    """updown.py:101-118."""
    s = 0
    while (prefix + f"{2 * s}.conv.weight") in sd:
        x = conv_relu_gn(sd, prefix + f"{2 * s}.", x, 2, 0)
        x = conv_relu_gn(sd, prefix + f"{2 * s + 1}.", x, 1, 0)
        s += 1
    return x


def encoder_points(sd, p, return_stages=False):
    """enc.py:115-133 — the per-point part (fc_pos, 5 blocks with 4 local pools, fc_c).

    p: (B,T,3) already scaled to [-.5,.5] (vqdif.py:36).  Returns c (B,T,32), cell (B,T).
    """
    u = normalize_3d(p)
    cell = cell_index(u, G)
    net = F.linear(p, sd["encoder.fc_pos.weight"], sd["encoder.fc_pos.bias"])
    net = resblock(sd, "encoder.blocks.0.", net)
    stages = [net]
    for i in range(1, 5):
        pooled = local_max_pool(net, cell)
        net = resblock(sd, f"encoder.blocks.{i}.", torch.cat([net, pooled], dim=2))
        stages.append(net)
    c = F.linear(net, sd["encoder.fc_c.weight"], sd["encoder.fc_c.bias"])
    if return_stages:
        return c, cell, u, stages
    return c, cell, u


def grid_mean(c, cell, ncell=G ** 3):
    """enc.py:66-75 scatter_mean into zeros -> (B,32,G,G,G) indexed [z][y][x]."""
    B, T, C = c.shape
    idx = cell[:, :, None].expand(B, T, C)
    s = torch.zeros(B, ncell, C, dtype=c.dtype).scatter_add_(1, idx, c)
    n = torch.zeros(B, ncell, C, dtype=c.dtype).scatter_add_(1, idx, torch.ones_like(c))
    m = s / n.clamp(min=1)
    return m.permute(0, 2, 1).reshape(B, C, G, G, G)


def occupancy_mask(u, R):
    """enc.py:85-91: mask[b, z, y, x] = True at trunc(u*R)."""
    B = u.shape[0]
    m = (u * R).long()
    mask = torch.zeros(B, R, R, R, dtype=torch.bool)
    b = torch.arange(B)[:, None].expand(B, u.shape[1])
    mask[b, m[..., 2], m[..., 1], m[..., 0]] = True
    return mask


def encode(sd, cloud):
    """VQDIF.encode (vqdif.py:35-37) = LocalPoolPointnet.forward(cloud/2) (enc.py:115-140)."""
    p = cloud / 2.0
    c, cell, u = encoder_points(sd, p)
    fea = downsampler(sd, grid_mean(c, cell))
    return fea, occupancy_mask(u, fea.shape[-1])


# ---- a9 / a10 ---------------------------------------------------------------
def quantize(sd, fea):
    """quantizer.py:31-64: argmax(-(|x|^2 - 2 x W^T + |w|^2)), first max on ties."""
    B, d = fea.shape[:2]
    W = sd["quantizer.embedding.weight"]
    x = fea.permute(0, 2, 3, 4, 1).contiguous().view(-1, d)
    dist = (x ** 2).sum(1, keepdim=True) - 2 * torch.mm(x, W.t()) + (W.t() ** 2).sum(0, keepdim=True)
    idx = torch.max(-dist, dim=1)[1]
    return idx.view(B, *fea.shape[2:]), dist


def get_code(sd, ind):
    """quantizer.py:19-30: W[ind] -> (B,d,R,R,R)."""
    return F.embedding(ind, sd["quantizer.embedding.weight"]).permute(0, 4, 1, 2, 3).contiguous()


def get_mode(idx):
    """models/common.py:20-23 — most frequent value, smallest on ties (== torch.mode on CPU)."""
    vals, counts = torch.unique(idx.reshape(-1), return_counts=True)
    return vals[torch.argmax(counts)]


def quantize_cloud(sd, cloud):
    """VQDIF.quantize_cloud (vqdif.py:50-58): whole-batch mode outside the mask."""
    fea, mask = encode(sd, cloud)
    ind, _ = quantize(sd, fea)
    mode = get_mode(ind)
    q = torch.zeros_like(ind) + mode
    q[mask] = ind[mask]
    return q, mode, dict(quant_ind=ind, grid_mask=mask, grid_feat=fea)


# ---- a21 / a22 ----------------------------------------------------------------
def single_gcr(sd, prefix, x):
    """unet3d.py:79-101 order 'gcr': GroupNorm(8) -> conv3^3(no bias, pad 1) -> ReLU."""
    x = F.group_norm(x, 8, sd[prefix + "groupnorm.weight"], sd[prefix + "groupnorm.bias"], 1e-5)
    return _relu(F.conv3d(x, sd[prefix + "conv.weight"], None, padding=1))


def double_conv(sd, prefix, x):
    return single_gcr(sd, prefix + "SingleConv2.", single_gcr(sd, prefix + "SingleConv1.", x))


def unet3d(sd, x, prefix="decoder.unet3d."):
    """unet3d.py:449-474 (3 levels, DoubleConv, nearest upsample + concat(skip, x), final 1x1+bias)."""
    e0 = double_conv(sd, prefix + "encoders.0.basic_module.", x)
    e1 = double_conv(sd, prefix + "encoders.1.basic_module.", _max_pool2(e0))
    e2 = double_conv(sd, prefix + "encoders.2.basic_module.", _max_pool2(e1))
    y = torch.cat([e1, F.interpolate(e2, size=e1.shape[2:], mode="nearest")], dim=1)
    y = double_conv(sd, prefix + "decoders.0.basic_module.", y)
    y = torch.cat([e0, F.interpolate(y, size=e0.shape[2:], mode="nearest")], dim=1)
    y = double_conv(sd, prefix + "decoders.1.basic_module.", y)
    return F.conv3d(y, sd[prefix + "final_conv.weight"], sd[prefix + "final_conv.bias"])


def upsampler(sd, x, prefix="decoder.upsampler.blocks."):
    """updown.py:119-132: per step nearest x2, two conv3^3 -> ReLU -> GN."""
    s = 0
    while (prefix + f"{3 * s + 1}.conv.weight") in sd:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = conv_relu_gn(sd, prefix + f"{3 * s + 1}.", x, 1, 1)
        x = conv_relu_gn(sd, prefix + f"{3 * s + 2}.", x, 1, 1)
        s += 1
    return x


def decoder_grid(sd, quant_feat):
    """dec.py:75-83: UNet3D then Upsampler -> (B,32,64,64,64)."""
    return upsampler(sd, unet3d(sd, quant_feat))


# ---- a23 ------------------------------------------------------------------------
def trilinear_sample(grid, p):
    """dec.py:62-68: grid_sample(align_corners=True, border) at 2*normalize(p)-1.

    grid (B,C,D,H,W) [z][y][x]; p (B,N,3) in [-.5,.5] -> (B,N,C).
    """
    u = normalize_3d(p)[:, :, None, None].to(grid.dtype)
    v = 2.0 * u - 1.0
    c = F.grid_sample(grid, v, padding_mode="border", align_corners=True, mode="bilinear")
    return c.squeeze(-1).squeeze(-1).transpose(1, 2)


def sdf_mlp(sd, p, c):
    """dec.py:88-100: fc_p, 5x(net += fc_c[i](c); ResnetBlockFC), fc_out(relu(net))."""
    net = F.linear(p, sd["decoder.fc_p.weight"], sd["decoder.fc_p.bias"])
    for i in range(5):
        net = net + F.linear(c, sd[f"decoder.fc_c.{i}.weight"], sd[f"decoder.fc_c.{i}.bias"])
        net = resblock(sd, f"decoder.blocks.{i}.", net)
    return F.linear(_relu(net), sd["decoder.fc_out.weight"], sd["decoder.fc_out.bias"])


def sdf_query(sd, grid, Xtg, chunk=1 << 18):
    """LocalDecoder implicit part on a precomputed (B,32,64,64,64) grid; Xtg in [-1,1]."""
    outs = []
    for s in range(0, Xtg.shape[1], chunk):
        p = (Xtg[:, s:s + chunk] / 2.0).to(grid.dtype)
        outs.append(sdf_mlp(sd, p, trilinear_sample(grid, p)))
    return torch.cat(outs, dim=1)


def decode_index(sd, ind, Xtg):
    """VQDIF.decode_index (vqdif.py:60-76): codes -> grid -> logits (B,N,1)."""
    return sdf_query(sd, decoder_grid(sd, get_code(sd, ind)), Xtg)


def make_grid(Q, dtype=np.float32):
    """nputil.makeGrid (xgutils/nputil.py:618-654) 'ij': f64 linspace(-1,1,Q)^3, x slowest."""
    ax = np.linspace(-1.0, 1.0, Q)
    g = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), axis=-1).reshape(-1, 3)
    return g.astype(dtype)
