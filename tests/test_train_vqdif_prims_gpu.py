"""GPU: every backward primitive of the VQDIF training step (csrc/train_vqdif.hip + the reused forward kernels) against
torch autograd of the same op in float64: GEMM / Linear weight gradient, conv3d forward / weight gradient / input
gradient (3^3 pad 1, 2^3 stride 2), GroupNorm, max-pool with ties, nearest upsample + concat, trilinear sampling,
the encoder's local max pool and scatter-mean.  Tolerance 2e-5 (max-normalised) - these are exact up to fp32 rounding."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def tr():
    from shapeformer_amd import weights as W
    from shapeformer_amd.train_vqdif import VQDIFTrainer
    assert torch.cuda.is_available()
    return VQDIFTrainer(W.make_state_dict(W.vqdif_spec(16)), device="cuda:0")


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-12))


def test_gemm_and_linear_weight_gradient(tr):
    from shapeformer_amd import _lib as L
    dev, lib = tr.dev, L.lib()
    torch.manual_seed(0)
    for M, N, K in ((512, 32, 32), (512, 64, 32), (700, 32, 64), (4096, 128, 128), (512, 32, 16)):
        x, Wt, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
        y = tr._gemm(x, Wt, b, None, M, N, K)
        assert rel(y, x.double() @ Wt.double().t() + b.double()) < TOL
        dy = torch.randn(M, N, device=dev)
        for nsplit in (1, 3):
            part = torch.empty(nsplit, N * K, device=dev)
            L.check(lib.sfmi_conv3d_wgrad_f32(L.ptr(dy), L.ptr(x), L.ptr(part), 1, 1, 1, M, K, N, 1, 1, 0, N, K, nsplit, L.stream_ptr()), "wgrad")
            assert rel(part.sum(0).view(N, K), dy.double().t() @ x.double()) < TOL


@pytest.mark.parametrize("B,D,Cin,Cout,KS,st,pad", [(1, 8, 32, 64, 3, 1, 1), (2, 8, 64, 32, 2, 2, 0), (1, 16, 128, 128, 3, 1, 1),
                                                   (2, 4, 768, 256, 3, 1, 1), (1, 8, 64, 64, 1, 1, 0)])
def test_conv3d_forward_weight_and_input_gradients(tr, B, D, Cin, Cout, KS, st, pad):
    from shapeformer_amd.train_vqdif import Tape
    dev = tr.dev
    torch.manual_seed(1)
    x = torch.randn(B, D, D, D, Cin, device=dev)
    w = torch.randn(Cout, Cin, KS, KS, KS, device=dev) * 0.1
    tr.p["tmpw"] = w.reshape(Cout, Cin, KS ** 3).permute(2, 0, 1).contiguous()
    tr.g["tmpw"] = torch.zeros_like(tr.p["tmpw"])
    tr.tape = Tape()
    y = tr.conv(x, "tmpw", B, D, Cin, Cout, KS, st, pad, True)
    xt, wt = x.double().permute(0, 4, 1, 2, 3).requires_grad_(True), w.double().requires_grad_(True)
    yt = F.relu(F.conv3d(xt, wt, None, stride=st, padding=pad))
    assert rel(y, yt.detach().permute(0, 2, 3, 4, 1)) < TOL
    # ReLU kink: keep the upstream gradient away from units that the two implementations could classify differently
    dy = torch.randn_like(y) * (yt.detach().permute(0, 2, 3, 4, 1).abs() > 1e-4).float() * (y.abs() > 1e-4).float()
    yt.backward(dy.double().permute(0, 4, 1, 2, 3))
    tr.tape.backward(y, dy, tr._add)
    assert rel(tr.g["tmpw"], wt.grad.reshape(Cout, Cin, KS ** 3).permute(2, 0, 1)) < TOL
    assert rel(tr.tape.grads[id(x)], xt.grad.permute(0, 2, 3, 4, 1)) < TOL


def test_groupnorm_pooling_upsampling(tr):
    from shapeformer_amd.train_vqdif import Tape
    dev = tr.dev
    torch.manual_seed(2)
    B, D, C = 2, 8, 64
    x = torch.randn(B, D, D, D, C, device=dev) * 2 + 0.5
    tr.p["g"], tr.p["b"] = torch.randn(C, device=dev), torch.randn(C, device=dev)
    tr.g["g"], tr.g["b"] = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    tr.tape = Tape()
    y = tr.groupnorm(x, "g", "b", B, D ** 3, C)
    dy = torch.randn_like(y)
    tr.tape.backward(y, dy, tr._add)
    xt, gt, bt = x.double().permute(0, 4, 1, 2, 3).requires_grad_(True), tr.p["g"].double().requires_grad_(True), tr.p["b"].double().requires_grad_(True)
    yt = F.group_norm(xt, 8, gt, bt, 1e-5)
    yt.backward(dy.double().permute(0, 4, 1, 2, 3))
    assert rel(y, yt.detach().permute(0, 2, 3, 4, 1)) < TOL and rel(tr.tape.grads[id(x)], xt.grad.permute(0, 2, 3, 4, 1)) < TOL
    assert rel(tr.g["g"], gt.grad) < TOL and rel(tr.g["b"], bt.grad) < TOL
    # max-pool of a post-ReLU tensor: whole windows of exact zeros (ties) must route like ATen (first element)
    x = torch.relu(torch.randn(B, 8, 8, 8, 32, device=dev))
    tr.tape = Tape()
    y = tr.maxpool(x, B, 4, 32)
    dy = torch.randn_like(y)
    tr.tape.backward(y, dy, tr._add)
    xt = x.double().permute(0, 4, 1, 2, 3).requires_grad_(True)
    yt = F.max_pool3d(xt, 2)
    yt.backward(dy.double().permute(0, 4, 1, 2, 3))
    assert rel(y, yt.detach().permute(0, 2, 3, 4, 1)) == 0 and rel(tr.tape.grads[id(x)], xt.grad.permute(0, 2, 3, 4, 1)) == 0
    # cat(skip, nearest x2 (low)) and plain nearest x2
    sk, lo = torch.randn(B, 8, 8, 8, 32, device=dev), torch.randn(B, 4, 4, 4, 64, device=dev)
    for Cs in (32, 0):
        tr.tape = Tape()
        y = tr.upcat(sk if Cs else None, lo, B, 8, Cs, 64)
        dy = torch.randn_like(y)
        tr.tape.backward(y, dy, tr._add)
        st_, lt = sk.double().permute(0, 4, 1, 2, 3).requires_grad_(True), lo.double().permute(0, 4, 1, 2, 3).requires_grad_(True)
        up = F.interpolate(lt, scale_factor=2, mode="nearest")
        yt = torch.cat([st_, up], 1) if Cs else up
        yt.backward(dy.double().permute(0, 4, 1, 2, 3))
        assert rel(y, yt.detach().permute(0, 2, 3, 4, 1)) == 0 and rel(tr.tape.grads[id(lo)], lt.grad.permute(0, 2, 3, 4, 1)) < TOL
        if Cs:
            assert rel(tr.tape.grads[id(sk)], st_.grad.permute(0, 2, 3, 4, 1)) == 0


def test_trilinear_and_encoder_pooling(tr):
    from oracle import vqdif_oracle as VO
    from shapeformer_amd import _lib as L
    dev, lib = tr.dev, L.lib()
    torch.manual_seed(3)
    grid = torch.randn(1, 64, 64, 64, 32, device=dev)
    Xtg = (torch.rand(1, 700, 3, device=dev) * 2.4 - 1.2).clamp(-1, 1)          # includes points on the border
    c = tr._f(700, 32)
    L.check(lib.sfmi_trilinear_cl_f32(L.ptr(Xtg), L.ptr(grid), L.ptr(c), 1, 700, 64, 32, L.stream_ptr()), "trilinear")
    gt = grid.cpu().permute(0, 4, 1, 2, 3).clone().requires_grad_(True)
    ct = VO.trilinear_sample(gt, Xtg.cpu() / 2)
    dc = torch.randn(700, 32, device=dev)
    ct.backward(dc.cpu()[None])
    acc = torch.zeros(1, 64 ** 3, 32, device=dev, dtype=torch.int64)
    L.check(lib.sfmi_trilinear_bwd_cl_f32(L.ptr(Xtg), L.ptr(dc), L.ptr(acc), 1, 700, 64, 32, L.stream_ptr()), "trilinear bwd")
    dg = tr._f(1, 64, 64, 64, 32)
    L.check(lib.sfmi_fixed_to_float_f32(L.ptr(acc), L.ptr(dg), acc.numel(), 0, L.stream_ptr()), "fixed_to_float")
    assert rel(c, ct.detach()[0]) < TOL and rel(dg, gt.grad.permute(0, 2, 3, 4, 1)) < TOL
    # cells, local max pool (+ concat) and scatter-mean, forward and backward
    B, T = 2, 3000
    cloud = (torch.rand(B, T, 3, device=dev) * 2 - 1) * 0.3                       # ~3 points per occupied cell
    cell = torch.empty(B, T, device=dev, dtype=torch.int32)
    L.check(lib.sfmi_cells_f32(L.ptr(cloud), L.ptr(cell), None, B, T, 64, L.stream_ptr()), "cells")
    cell_o = VO.cell_index(VO.normalize_3d(cloud.cpu() / 2), 64)
    assert bool((cell.cpu().long() == cell_o).all())
    net = torch.randn(B * T, 32, device=dev)
    keys = torch.empty(B, 64 ** 3, 32, device=dev, dtype=torch.int32)
    keys.view(torch.uint8).fill_(0x80)
    cat = tr._f(B * T, 64)
    cat[:, :32] = net
    L.check(lib.sfmi_cell_max_f32(L.ptr(net), L.ptr(cell), L.ptr(keys), L.ptr(cat), B, T, 64 ** 3, 32, 64, 32, L.stream_ptr()), "cell_max")
    nt = net.cpu().view(B, T, 32).requires_grad_(True)
    catt = torch.cat([nt, VO.local_max_pool(nt, cell_o)], 2)
    dcat = torch.randn(B * T, 64, device=dev)
    catt.backward(dcat.cpu().view(B, T, 64))
    acc = torch.zeros(B, 64 ** 3, 32, device=dev, dtype=torch.int64)
    L.check(lib.sfmi_cell_scatter_add_f32(L.ptr(dcat), L.ptr(cell), L.ptr(acc), None, B, T, 64 ** 3, 32, 64, 32, L.stream_ptr()), "scatter")
    dnet = dcat[:, :32].contiguous()
    L.check(lib.sfmi_cell_max_bwd_f32(L.ptr(net), L.ptr(keys), L.ptr(acc), L.ptr(cell), L.ptr(dnet), B, T, 64 ** 3, 32, 32, 1, L.stream_ptr()), "max bwd")
    assert rel(cat, catt.detach().view(-1, 64)) == 0 and rel(dnet, nt.grad.view(-1, 32)) < TOL
    cc = torch.randn(B * T, 32, device=dev)
    acc = torch.zeros(B, 64 ** 3, 32, device=dev, dtype=torch.int64)
    cnt = torch.zeros(B, 64 ** 3, device=dev, dtype=torch.int32)
    L.check(lib.sfmi_cell_scatter_add_f32(L.ptr(cc), L.ptr(cell), L.ptr(acc), L.ptr(cnt), B, T, 64 ** 3, 32, 32, 0, L.stream_ptr()), "scatter")
    grid = tr._f(B, 64, 64, 64, 32)
    L.check(lib.sfmi_cell_mean_f32(L.ptr(acc), L.ptr(cnt), L.ptr(grid), B, 64 ** 3, 32, L.stream_ptr()), "mean")
    ct_ = cc.cpu().view(B, T, 32).requires_grad_(True)
    gm = VO.grid_mean(ct_, cell_o)
    dgrid = torch.randn(B, 64, 64, 64, 32, device=dev)
    gm.backward(dgrid.cpu().permute(0, 4, 1, 2, 3))
    dcc = tr._f(B * T, 32)
    L.check(lib.sfmi_cell_mean_bwd_f32(L.ptr(dgrid), L.ptr(cnt), L.ptr(cell), L.ptr(dcc), B, T, 64 ** 3, 32, L.stream_ptr()), "mean bwd")
    assert rel(grid, gm.detach().permute(0, 2, 3, 4, 1)) < TOL and rel(dcc, ct_.grad.view(-1, 32)) < TOL


def test_bce_and_sdf_head_chain(tr):
    """Loss kernel + the whole implicit-decoder MLP chain against float64 autograd, on inputs with a margin on every ReLU."""
    from shapeformer_amd import _lib as L
    from shapeformer_amd.train_vqdif import Tape
    dev, lib = tr.dev, L.lib()
    torch.manual_seed(4)
    x, t = torch.randn(4000, device=dev) * 4, (torch.rand(4000, device=dev) > 0.5).float()
    rows, dx = tr._f(4000), tr._f(4000)
    L.check(lib.sfmi_bce_logits_f32(L.ptr(x), L.ptr(t), L.ptr(rows), L.ptr(dx), 4000, 1.0 / 4000, L.stream_ptr()), "bce")
    xt = x.double().requires_grad_(True)
    lt = F.binary_cross_entropy_with_logits(xt, t.double())
    lt.backward()
    assert abs(float(rows.double().mean()) - float(lt.detach())) < 1e-6 and rel(dx, xt.grad) < TOL
