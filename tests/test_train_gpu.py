"""GPU parity: training step of CondTupleGPT (forward loss, every parameter gradient, AdamW update) through the C ABI
vs the CPU oracle's autograd (oracle/gpt_oracle.py is pinned to the reference forward/loss)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _setup(dev):
    from oracle import gpt_oracle as GO, vqdif_oracle as VO
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=96)
    sd = W.make_state_dict(W.gpt_spec(**kw))
    cfg = GO.GPTCfg(n_embd=128, n_head=2, n_layers=(2, 1), block_size=96)
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    t = np.load(os.path.join(G, "gpt_tiny.npz"))
    c, z = torch.from_numpy(t["c_idx"]), torch.from_numpy(t["z_idx"])   # (2,24,2), (2,40,2) reference token tensors
    return sd, cfg, g, c, z


def _oracle_grads(sd, cfg, c, z, dropout=None):
    from oracle import gpt_oracle as GO, tokens_oracle as TO, vqdif_oracle as VO
    sdt = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
    extra = torch.from_numpy(TO.extra_indices_AR_N(c.numpy(), z.numpy(), 4096))
    loss = GO.training_loss(sdt, cfg, c, z, extra, dropout=dropout)
    loss.backward()
    return loss.item(), {k: v.grad for k, v in sdt.items()}, sdt


def _map(trainer, cfg):
    """trainer grad name -> oracle state-dict key(s)."""
    m = {}
    li = 0
    for s, nl in enumerate(cfg.n_layers):
        for n in range(nl):
            p, q = f"L{li}.", f"blocks.{s}.{n}."
            m[p + "ln1.w"], m[p + "ln1.b"] = [q + "ln1.weight"], [q + "ln1.bias"]
            m[p + "ln2.w"], m[p + "ln2.b"] = [q + "ln2.weight"], [q + "ln2.bias"]
            m[p + "wqkv"] = [q + f"attn.{x}.weight" for x in ("query", "key", "value")]
            m[p + "bqkv"] = [q + f"attn.{x}.bias" for x in ("query", "key", "value")]
            m[p + "wproj"], m[p + "bproj"] = [q + "attn.proj.weight"], [q + "attn.proj.bias"]
            m[p + "wfc1"], m[p + "bfc1"] = [q + "mlp.0.weight"], [q + "mlp.0.bias"]
            m[p + "wfc2"], m[p + "bfc2"] = [q + "mlp.2.weight"], [q + "mlp.2.bias"]
            li += 1
    for s in range(2):
        m[f"head{s}.ln.w"], m[f"head{s}.ln.b"], m[f"head{s}.w"] = [f"heads.{s}.0.weight"], [f"heads.{s}.0.bias"], [f"heads.{s}.1.weight"]
    m["E0"], m["E1"], m["Ex"] = ["tok_embs.0.weight"], ["tok_embs.1.weight"], ["extra_tok_embs.0.weight"]
    m["pos_emb"], m["cond_pos_emb"] = ["pos_emb"], ["cond_pos_emb"]
    return m


def test_loss_and_every_gradient_vs_oracle_autograd(dev):
    from shapeformer_amd.train import GPTTrainer
    sd, cfg, g, c, z = _setup(dev)
    tr = GPTTrainer(g)
    tr.debug_poison_grads = True      # every gradient outside the embedding tables must be WRITTEN by the backward pass (the step only zeroes those)
    loss = tr.loss_and_grad(c, z).item()
    want_loss, og, _ = _oracle_grads(sd, cfg, c, z)
    assert abs(loss - want_loss) < 1e-5 * max(1.0, abs(want_loss))
    worst = 0.0
    for name, keys in _map(tr, cfg).items():
        want = torch.cat([og[k].reshape(-1, og[k].shape[-1]) if og[k].dim() > 1 else og[k] for k in keys], 0)
        got = tr.grad[name].cpu().reshape(want.shape)
        scale = want.abs().max().item() + 1e-12
        err = (got - want).abs().max().item() / scale
        worst = max(worst, err)
        assert err < 2e-3, (name, err, scale)   # fp32 reassociation across ~1e3-term reductions; typical 1e-5
    print(f"worst relative gradient error {worst:.2e}")
    # bit-reproducible gradients (fixed-order reductions / fixed-point atomics)
    g1 = tr.flat_grad.clone()
    tr.loss_and_grad(c, z)
    assert torch.equal(g1, tr.flat_grad)


def test_train_mode_dropout_loss_and_every_gradient_vs_oracle_autograd(dev):
    """The three dropouts of the reference (mingpt.py:62-63,85,90,105,218,292: embeddings of both stages, attention
    probabilities, proj / MLP outputs) in TRAIN mode: fused counter-hash masks on the HIP side, the same masks as explicit
    multipliers in the oracle's autograd.  p = 0.1 (the YAML's 0.01 would barely move the numbers)."""
    from shapeformer_amd.train import GPTTrainer
    sd, cfg, g, c, z = _setup(dev)
    pd = (0.1, 0.15, 0.2)
    tr = GPTTrainer(g, pdrop=pd)
    l_eval = tr.loss_and_grad(c, z).item()
    loss = tr.loss_and_grad(c, z, dropout_key="k7").item()
    want_loss, og, _ = _oracle_grads(sd, cfg, c, z, dropout=dict(key="k7", p=pd))
    assert abs(loss - want_loss) < 1e-5 * max(1.0, abs(want_loss)) and abs(loss - l_eval) > 1e-3      # the masks are live
    worst = 0.0
    for name, keys in _map(tr, cfg).items():
        want = torch.cat([og[k].reshape(-1, og[k].shape[-1]) if og[k].dim() > 1 else og[k] for k in keys], 0)
        got = tr.grad[name].cpu().reshape(want.shape)
        err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
        worst = max(worst, err)
        assert err < 2e-3, (name, err)
    print(f"train mode (dropout {pd}): worst relative gradient error {worst:.2e}")
    # another key -> other masks; training_step draws a fresh key per step and still trains
    assert abs(tr.loss_and_grad(c, z, dropout_key="k8").item() - loss) > 1e-4
    tr2 = GPTTrainer(g, lr=1e-3, pdrop=(0.01, 0.01, 0.01))
    losses = [tr2.training_step(c, z).item() for _ in range(6)]
    assert losses[-1] < losses[0] - 0.05, losses


def test_adamw_step_matches_torch_optim_and_training_reduces_loss(dev):
    from shapeformer_amd.train import GPTTrainer
    sd, cfg, g, c, z = _setup(dev)
    tr = GPTTrainer(g, lr=1e-3)
    want_loss, og, sdt = _oracle_grads(sd, cfg, c, z)
    m = _map(tr, cfg)
    decay = [sdt[k] for n, ks in m.items() for k in ks if n.split(".")[-1] in ("wqkv", "wproj", "wfc1", "wfc2", "w") and "ln" not in n]
    dset = {id(p) for p in decay}
    nodecay = [p for p in sdt.values() if id(p) not in dset]
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    l0 = tr.training_step(c, z).item()
    opt.step()
    for name, keys in m.items():
        want = torch.cat([sdt[k].detach().reshape(-1, sdt[k].shape[-1]) if sdt[k].dim() > 1 else sdt[k].detach() for k in keys], 0)
        got = dict((n, t) for n, t, _ in tr.params)[name].cpu().reshape(want.shape)
        gref = torch.cat([og[k].reshape(-1, og[k].shape[-1]) if og[k].dim() > 1 else og[k] for k in keys], 0)
        # Adam normalises by |g|: where the true gradient is zero (e.g. the key bias, softmax is shift invariant) the
        # update direction is the sign of round-off noise -> compare only elements with a resolvable gradient
        sig = gref.abs() > 1e-3 * max(gref.abs().max().item(), 1e-12)
        assert ((got - want).abs() * sig).max().item() < 2e-5, name     # one AdamW step of size lr=1e-3
        assert (got - want).abs().max().item() < 2.1e-3, name           # noise elements move by at most 2*lr
    losses = [l0] + [tr.training_step(c, z).item() for _ in range(5)]
    assert losses[-1] < losses[0] - 0.05, losses                # the step actually trains
    # decode-path weights follow the raw weights: sampling after training uses the updated model
    out = g.sample(c[:, :24].to(torch.int32), torch.tensor([24, 24], dtype=torch.int32), max_steps=4, stop_early=False)
    assert out["samples"].shape == (2, 4, 2)


def test_library_gemm_binding_and_large_row_training_path(dev, monkeypatch):
    """csrc/blas.hip: the lazily bound rocBLAS sgemm (row-major wrapper, all transpose forms, fused epilogue) against float64,
    and the training step at M = B*L >= 1024 rows, where the trainer routes its plain GEMMs through it."""
    from oracle import gpt_oracle as GO
    from shapeformer_amd import _lib as L, weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.train import GPTTrainer
    lib = L.lib()
    assert lib.sfmi_blas_available() == 1, "rocBLAS (librocblas.so.5) must be bindable on the GPU box"
    torch.manual_seed(0)
    rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-12))
    M, N, K = 1500, 96, 200
    A, B_ = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    for tA in (0, 1):
        for tB in (0, 1):
            As, Bs = (A.t().contiguous() if tA else A), (B_.t().contiguous() if tB else B_)
            C = torch.randn(M, N, device=dev)
            want = 0.5 * (A.double() @ B_.double()) + 2.0 * C.double()
            L.check(lib.sfmi_sgemm_f32(tA, tB, M, N, K, 0.5, L.ptr(As), As.shape[1], L.ptr(Bs), Bs.shape[1], 2.0, L.ptr(C), N,
                                       L.stream_ptr()), "sgemm")
            assert rel(C, want) < 1e-5, (tA, tB)
    x, Wt, b, r = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev), torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev)
    L.check(lib.sfmi_gemm_blas_f32(L.ptr(x), L.ptr(Wt), L.ptr(b), None, L.ptr(y), M, N, K, 2, L.stream_ptr()), "gemm_blas gelu")
    assert rel(y, torch.nn.functional.gelu(x.double() @ Wt.double().t() + b.double())) < 1e-5
    y = r.clone()
    L.check(lib.sfmi_gemm_blas_f32(L.ptr(x), L.ptr(Wt), L.ptr(b), L.ptr(y), L.ptr(y), M, N, K, 0, L.stream_ptr()), "gemm_blas inplace resid")
    assert rel(y, x.double() @ Wt.double().t() + b.double() + r.double()) < 1e-5
    # training step with 4 x 349 = 1396 rows
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=400)
    sd = W.make_state_dict(W.gpt_spec(**kw))
    cfg = GO.GPTCfg(n_embd=128, n_head=2, n_layers=(2, 1), block_size=400)
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=400, device=dev)
    rs = np.random.RandomState(5)

    def rows(Lr):
        out = np.full((4, Lr, 2), 4096, np.int64)
        for bb in range(4):
            n = rs.randint(Lr // 2, Lr)
            out[bb, :n, 0] = np.sort(rs.choice(4096, n, replace=False))
            out[bb, :n, 1] = rs.randint(0, 4096, n)
        return torch.from_numpy(out)
    c, z = rows(150), rows(200)
    want_loss, og, _ = _oracle_grads(sd, cfg, c, z)
    # default: every GEMM of the step on csrc/sgemm.hip (forward, dX = dY W, dW = dY^T X with split-K); SFMI_ROCBLAS=1: library
    for use_lib in (False, True):
        if use_lib:
            monkeypatch.setenv("SFMI_ROCBLAS", "1")
        tr = GPTTrainer(g)
        assert tr._blas() == use_lib
        loss = tr.loss_and_grad(c, z).item()
        assert abs(loss - want_loss) < 1e-5 * max(1.0, abs(want_loss))
        for name, keys in _map(tr, cfg).items():
            want = torch.cat([og[k].reshape(-1, og[k].shape[-1]) if og[k].dim() > 1 else og[k] for k in keys], 0)
            got = tr.grad[name].cpu().reshape(want.shape)
            err = (got - want).abs().max().item() / (want.abs().max().item() + 1e-12)
            assert err < 2e-3, (name, use_lib, err)


@pytest.mark.parametrize("B,L,H", [(1, 499, 16), (2, 150, 2), (3, 64, 4), (1, 33, 2), (4, 300, 16)])
def test_attention_backward_fused_launch_from_the_forward_lse(dev, B, L, H):
    """Round 5: the training forward keeps the rows' log-sum-exps (sfmi_gpt_attn_prefill_lse_f32) and the backward is a row-sum launch
    + ONE launch that runs the dQ and the dK / dV blocks side by side (sfmi_attn_bwd_lse_f32).  Against torch autograd of the same
    causal attention (mingpt.py:73-91, fp64 on the CPU), and against the form that recomputes the log-sum-exps (sfmi_attn_bwd_f32)."""
    from shapeformer_amd import _lib as L_
    lib = L_.lib()
    D = 64 * H
    g = torch.Generator().manual_seed(B * 1000 + L)
    qkv = torch.randn(B * L, 3 * D, generator=g)
    dy = torch.randn(B * L, D, generator=g)
    # reference
    x = qkv.double().clone().requires_grad_(True)
    q, k, v = (x[:, i * D:(i + 1) * D].view(B, L, H, 64).transpose(1, 2) for i in range(3))
    att = (q @ k.transpose(-1, -2)) / 8.0
    att = att.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf")).softmax(-1)
    yref = (att @ v).transpose(1, 2).reshape(B * L, D)
    yref.backward(dy.double())
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2) / 8.0).masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool)), float("-inf")), -1)
    # device
    qd, dyd = qkv.to(dev), dy.to(dev)
    Lmax = L + 1
    kv = torch.empty(2, B, Lmax, D, device=dev)
    nval = torch.full((B,), L, device=dev, dtype=torch.int32)
    y, lse = torch.empty(B * L, D, device=dev), torch.empty(B, H, L, device=dev)
    L_.check(lib.sfmi_gpt_attn_prefill_lse_f32(L_.ptr(qd), L_.ptr(kv[0]), L_.ptr(kv[1]), L_.ptr(nval), L_.ptr(y), B, L, D, H, Lmax, None, 0.0, 0,
                                               L_.ptr(lse), L_.stream_ptr()), "attn fwd")
    assert float((y.cpu().double() - yref.detach()).abs().max()) < 2e-5
    assert float((lse.cpu().double() - lse_ref.detach()).abs().max()) < 2e-5
    # the small-launch forward (32-row tiles x two key-block groups, merged online-softmax states): same y / lse; with attention
    # dropout the same mask as the prefill kernel
    if B * H * ((L + 63) // 64) <= 128:
        y2, lse2 = torch.full_like(y, float("nan")), torch.full_like(lse, float("nan"))
        L_.check(lib.sfmi_attn_train_fwd_small_f32(L_.ptr(qd), L_.ptr(y2), L_.ptr(lse2), B, L, D, H, 0.0, 0, L_.stream_ptr()), "attn fwd small")
        assert float((y2.cpu().double() - yref.detach()).abs().max()) < 2e-5 and float((lse2 - lse).abs().max()) < 1e-5
        y3, y4 = torch.empty_like(y), torch.empty_like(y)
        L_.check(lib.sfmi_attn_train_fwd_small_f32(L_.ptr(qd), L_.ptr(y3), L_.ptr(lse2), B, L, D, H, 0.2, 7, L_.stream_ptr()), "attn fwd small")
        L_.check(lib.sfmi_gpt_attn_prefill_lse_f32(L_.ptr(qd), L_.ptr(kv[0]), L_.ptr(kv[1]), L_.ptr(nval), L_.ptr(y4), B, L, D, H, Lmax, None, 0.2, 7,
                                                   None, L_.stream_ptr()), "attn fwd")
        assert float((y3 - y4).abs().max()) < 2e-5 and float((y3 - y).abs().max()) > 1e-3      # the masks are live and equal
    else:
        assert lib.sfmi_attn_train_fwd_small_f32(L_.ptr(qd), L_.ptr(y), L_.ptr(lse), B, L, D, H, 0.0, 0, L_.stream_ptr()) == -1
    delta, dq1 = torch.empty(B, H, L, device=dev), torch.full((B * L, 3 * D), float("nan"), device=dev)
    L_.check(lib.sfmi_attn_bwd_lse_f32(L_.ptr(qd), L_.ptr(y), L_.ptr(dyd), L_.ptr(lse), L_.ptr(delta), L_.ptr(dq1), B, L, D, H, 0.0, 0, L_.stream_ptr()), "bwd lse")
    scale = float(x.grad.abs().max())
    assert float((dq1.cpu().double() - x.grad).abs().max()) < 3e-5 * scale
    scratch, dq2 = torch.empty(2, B, H, L, device=dev), torch.full((B * L, 3 * D), float("nan"), device=dev)
    L_.check(lib.sfmi_attn_bwd_f32(L_.ptr(qd), L_.ptr(y), L_.ptr(dyd), L_.ptr(scratch), L_.ptr(dq2), B, L, D, H, 0.0, 0, L_.stream_ptr()), "bwd stats")
    assert float((dq1 - dq2).abs().max()) < 2e-5 * scale
    assert float((scratch[0] - lse).abs().max()) < 1e-5 and float((scratch[1] - delta).abs().max()) < 1e-4 * float(delta.abs().max())
    # with attention dropout the two forms still agree (same counter-hash mask in both halves)
    L_.check(lib.sfmi_attn_bwd_lse_f32(L_.ptr(qd), L_.ptr(y), L_.ptr(dyd), L_.ptr(lse), L_.ptr(delta), L_.ptr(dq1), B, L, D, H, 0.1, 99, L_.stream_ptr()), "bwd lse")
    L_.check(lib.sfmi_attn_bwd_f32(L_.ptr(qd), L_.ptr(y), L_.ptr(dyd), L_.ptr(scratch), L_.ptr(dq2), B, L, D, H, 0.1, 99, L_.stream_ptr()), "bwd stats")
    assert float((dq1 - dq2).abs().max()) < 2e-5 * scale and bool(torch.isfinite(dq1).all())


def test_training_step_gemm_forms_agree(dev):
    """The step on the work-balanced GEMM with fused GELU epilogues (gemm="sk", the default) and on the round 2-4 form (one workgroup
    per tile + split-K reduce + separate GELU launches, gemm="tile") give the same loss and gradients to fp32 rounding."""
    from shapeformer_amd.train import GPTTrainer
    sd, cfg, g, c, z = _setup(dev)
    ta, tb = GPTTrainer(g, gemm="sk"), GPTTrainer(g, gemm="tile")
    la, lb = ta.loss_and_grad(c, z, dropout_key="k"), tb.loss_and_grad(c, z, dropout_key="k")
    assert abs(float(la) - float(lb)) < 1e-5
    for name in ta.grad:
        a, b = ta.grad[name], tb.grad[name]
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7, name


@pytest.mark.parametrize("pdrop", [(0.0, 0.0, 0.0), (0.1, 0.15, 0.2)])
def test_captured_training_step_is_bit_identical_to_the_eager_step(dev, pdrop):
    """GPTTrainer(graph=True): forward + backward + per-bucket AdamW of a step replayed as ONE hipGraph (three streams; token tensors,
    dropout seeds and AdamW's bias corrections read from device memory).  Five steps on two alternating batches - the first of each shape
    eager, the second captured, then replays, the learning rate changed on the way - must give the losses AND the weights / moments of the
    eager trainer bit for bit, with dropout off and on (the per-step masks, bias corrections and learning rate come from the step words)."""
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.train import GPTTrainer
    t = np.load(os.path.join(G, "gpt_tiny.npz"))
    c, z = torch.from_numpy(t["c_idx"]), torch.from_numpy(t["z_idx"])
    batches = [(c, z), (c[:1], z[:1]), (c, z), (c.flip(0), z.flip(0)), (c[:1], z[:1]), (c, z), (c[:1], z[:1])]
    out = []
    for graph in (False, True):
        kw = dict(n_embd=128, n_layers=(2, 1), block_size=96)
        g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
        tr = GPTTrainer(g, lr=1e-3, pdrop=pdrop, graph=graph)
        losses = []
        for i, (cc, zz) in enumerate(batches):
            if i == 5:
                tr.lr = 0.5e-3       # a scheduled learning rate: a step word of the captured step, not a reason to capture again
            losses.append(float(tr.training_step(cc, zz).item()))
        if graph:
            assert len(tr._graphs) == 2 and tr.step_count == len(batches)
            assert (len(next(iter(tr._graphs.values()))["sites"]) > 0) == any(pdrop)
        out.append((losses, [p.detach().cpu().clone() for _, p, _ in tr.params], tr.flat_m.cpu().clone(), tr.flat_v.cpu().clone()))
    (l0, w0, m0, v0), (l1, w1, m1, v1) = out
    assert l0 == l1, (l0, l1)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1)) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert l0[-2] < l0[0]
