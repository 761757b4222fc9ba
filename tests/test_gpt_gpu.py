"""GPU parity: KV-cached CondTupleGPT sampling (prefill + decode step + fused sampler, all through the
C ABI) vs the CPU oracle, which is itself pinned to the reference's sample_indices (tests/golden)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
LOGIT_TOL = 1e-3  # SURVEY App.B: transformer logits f32 |d| <= 1e-3


def _tiny():
    from oracle import gpt_oracle as GO, vqdif_oracle as VO
    from shapeformer_amd import weights as W
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=96)
    sd = W.make_state_dict(W.gpt_spec(**kw))
    cfg = GO.GPTCfg(n_embd=128, n_head=2, n_layers=(2, 1), block_size=96)
    return sd, VO.to_torch_sd(sd), cfg


def _cond_rows():
    z = np.load(os.path.join(G, "vqdif16_small.npz"))
    tok = z["tokens"].astype(np.int64)  # (2, L, 2) reference tokens of the car / armchair clouds
    rows = [tok[0, :23], tok[1, :9], tok[0, 40:41]]
    Lc = [len(r) + 1 for r in rows]
    c = np.full((3, max(Lc), 2), 4096, np.int64)
    for b, r in enumerate(rows):
        c[b, :len(r)] = r
    return c, np.array(Lc, np.int32)


@pytest.mark.parametrize("use_graph", [False, True])
def test_tiny_ragged_sampling_vs_oracle(dev, use_graph):
    from oracle import gpt_oracle as GO
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    c, Lc = _cond_rows()
    steps, seed = 24, 3
    out = g.sample(torch.from_numpy(c), torch.from_numpy(Lc), max_steps=steps, seed=seed, stop_early=False,
                   use_graph=use_graph, return_logits=True)
    got, hist = out["samples"].numpy(), [h.numpy() for h in out["logits_history"]]
    assert got.shape == (3, steps, 2)
    u = GO.uniforms(seed, steps, 3)
    n_tok_mismatch = 0
    for b in range(3):
        cb = torch.from_numpy(c[b:b + 1, :Lc[b]])
        # (i) teacher-forced stepwise parity: feed OUR tokens to the oracle, compare every step's masked logits
        _, oh, _ = GO.sample_indices(sd_t, cfg, cb, steps, u[:, :, b:b + 1], use_cache=False, stop_early=False,
                                     force_tokens=got[b:b + 1], best_in_first=(b == 0))
        for i in range(2):
            a, r = hist[i][b], oh[i][0]
            fin = np.isfinite(r)
            assert np.array_equal(np.isfinite(a), fin), f"mask differs row {b} tuple {i}"
            assert np.abs(a[fin] - r[fin]).max() < LOGIT_TOL
        # (ii) free-running: same uniforms -> same tokens (greedy row 0 and stochastic rows)
        ot, _, _ = GO.sample_indices(sd_t, cfg, cb, steps, u[:, :, b:b + 1], use_cache=True, stop_early=False,
                                     best_in_first=(b == 0), return_logits=False)
        n_tok_mismatch += int((ot[0] != got[b]).any(-1).sum())
        lp = GO.compute_log_probs(got[b:b + 1], [h[b:b + 1] for h in hist])
        assert np.allclose(out["log_prob"][b].numpy(), lp[0], atol=2e-3)
    assert n_tok_mismatch == 0


def test_early_stop_and_end_token_rows(dev):
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    c, Lc = _cond_rows()
    # force every row to emit the end token at step 2: afterwards the masker must keep emitting end tokens
    ft = np.zeros((3, 8, 2), np.int64)
    ft[:, 0] = [5, 7]; ft[:, 1] = [4096, 4096]; ft[:, 2:] = [4096, 4096]
    out = g.sample(torch.from_numpy(c), torch.from_numpy(Lc), max_steps=8, stop_early=False, force_tokens=ft,
                   return_logits=True)
    h0, h1 = out["logits_history"]
    # step 2 (after an end token): position logits all -inf except the end token; value forced to end (1.0)
    assert torch.isinf(h0[:, 2, :4096]).all() and torch.isfinite(h0[:, 2, 4096]).all()
    assert (h1[:, 2, 4096] == 1.0).all() and torch.isinf(h1[:, 2, :4096]).all()
    # block_size cap: steps are clipped so that Lc + steps <= block_size
    out = g.sample(torch.from_numpy(c), torch.from_numpy(Lc), max_steps=500, stop_early=False)
    assert out["steps"] == 96 - int(Lc.max())


def test_full_size_probe_vs_reference_logits(dev):
    """Full 20+4 layer d=1024 model: logits at 5 positions produced by the REFERENCE forward (fixture)."""
    from shapeformer_amd.gpt import CondTupleGPT
    z = np.load(os.path.join(G, "gpt_full_probe.npz"))
    cz, L_c = z["cz"], int(z["L_c"])
    g = CondTupleGPT(device=dev)  # hash weights generated on the device (bit-identical to the numpy generator)
    c = torch.from_numpy(cz[:, :L_c])
    force = cz[:, L_c:]
    steps = force.shape[1]
    out = g.sample(c, torch.tensor([L_c], dtype=torch.int32), max_steps=steps, stop_early=False,
                   force_tokens=force, return_logits=True, mask_invalid=False, mask_invalid_completion=False)
    h0, h1 = (h.numpy()[0] for h in out["logits_history"])
    for k, t in enumerate(z["pos_sel"]):
        j = int(t) - (L_c - 1)   # reference output index t predicts generated token j
        if j < 0 or j >= steps:
            continue
        assert np.abs(h0[j] - z["logits0"][k]).max() < LOGIT_TOL
        fin = np.isfinite(h1[j])   # tuple-1 rows are only masked when pos == end
        assert np.abs(h1[j][fin] - z["logits1"][k][fin]).max() < LOGIT_TOL


def test_sampler_kernel_vs_oracle_filter(dev):
    """Fused masker/top-k/top-p/inverse-CDF kernel on random logits vs oracle/tokens_oracle.py."""
    from oracle import tokens_oracle as TO
    from shapeformer_amd import _lib as L, weights as W
    B, V, Vpad, Lmax = 8, 4097, 4128, 64
    rng = np.random.RandomState(1)
    logits = (rng.randn(B, V) * 3).astype(np.float32)
    part = np.zeros((1, B, Vpad), np.float32); part[0, :, :V] = logits
    seq = np.zeros((B, Lmax, 2), np.int32)
    Lc = np.full(B, 4, np.int32)
    for b in range(B):
        seq[b, :4, 0] = [10 * b + 1, 500 + b, 2000 + b, 4096]; seq[b, :4, 1] = [1, 2, 3, 4096]
        seq[b, 4] = [300 + 40 * b, 7]          # one generated token -> j = 1 when sampling position 5
    ln = np.full(B, 5, np.int32)
    bad = 0
    for (k, p, T) in [(100, 0.4, 1.0), (300, 0.9, 1.0), (50, 0.0, 0.7), (0, 0.8, 1.3)]:
        d = lambda a: torch.from_numpy(a.copy()).to(dev)
        dseq, dlen, dLc, dpart = d(seq), d(ln), d(Lc), d(part)
        hist = torch.empty(B, 4, V, device=dev)
        seed = W._fnv1a32("sample-uniforms-9")
        L.check(L.lib().sfmi_gpt_sample_f32(L.ptr(dpart), L.ptr(dseq), L.ptr(dlen), L.ptr(dLc), None, L.ptr(hist), None,
                                            None, None, None, None, None, 0, 1, B,
                                            V, Vpad, Lmax, 0, 4096, 4096, k, p, T, 0, 1, 1, 4, seed, None, 0, 0, B, 0, L.stream_ptr()), "sample")
        got = dseq.cpu().numpy()[:, 5, 0]
        u = W.hash_unit("sample-uniforms-9", 4 * 2 * B).reshape(4, 2, B)
        idx = np.concatenate([seq[:, :5], np.zeros((B, 1, 2), np.int32)], 1)
        ml = TO.sampling_masker(logits, idx, 4, 1, 0, (4096, 4096), True, True)
        assert np.array_equal(hist.cpu().numpy()[:, 1], ml)
        for b in range(B):
            f = TO.filter_sampling_logits(ml[b], k, p, T)
            want = TO.sample_filtered(f, u[1, 0, b])
            bad += int(want != got[b])
    assert bad == 0


def test_teacher_forced_forward_and_loss_vs_reference_vectors(dev):
    """CondTupleGPT.forward (mingpt.py:311-319) + the training loss (shapeformer.py:132-140), forward only:
    full 20+4-layer model vs logits produced by the REFERENCE (fixture), tiny model vs the oracle."""
    from oracle import gpt_oracle as GO
    from shapeformer_amd.gpt import CondTupleGPT
    z = np.load(os.path.join(G, "gpt_full_probe.npz"))
    cz, ex, L_c = torch.from_numpy(z["cz"]), torch.from_numpy(z["extra"]), int(z["L_c"])
    g = CondTupleGPT(device=dev)
    for extra in (ex[:, :-1], None):     # explicit extra index, and the built-in AR_N rule (representers.py:188-196)
        lg = g.forward(cz[:, :-1], extra, L_c, cz[:, 1:])
        for k, t in enumerate(z["pos_sel"]):
            assert np.abs(lg[0][0, t].cpu().numpy() - z["logits0"][k]).max() < LOGIT_TOL
            assert np.abs(lg[1][0, t].cpu().numpy() - z["logits1"][k]).max() < LOGIT_TOL
    del g
    sd, sd_t, cfg = _tiny()
    gt = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    t = np.load(os.path.join(G, "gpt_tiny.npz"))
    c, zz, exx = (torch.from_numpy(t[k]) for k in ("c_idx", "z_idx", "extra"))
    want = GO.training_loss(sd_t, cfg, c, zz, exx).item()
    got = gt.training_loss(c, zz, exx).item()
    assert abs(got - want) < 1e-4 * max(1.0, abs(want))


def test_microbatched_two_stream_decode_is_bit_identical(dev):
    """sample_microbatched (two hipGraph chains on two HIP streams) must give exactly the tokens of the single-batch
    run: uniforms and the greedy row are indexed by global row."""
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    c3, Lc3 = _cond_rows()
    c = np.concatenate([c3, c3[::-1]], 0)
    Lc = np.concatenate([Lc3, Lc3[::-1]], 0)
    ct, Lt = torch.from_numpy(c).to(dev, torch.int32), torch.from_numpy(Lc).to(dev)
    a = g.sample(ct, Lt, max_steps=20, seed=11, stop_early=False, to_host=False)
    seq_a, len_a, lp_a = a["state"]["seq"].clone(), a["state"]["len"].clone(), a["state"]["logp"].clone()
    for nm in (2, 3):
        b = g.sample_microbatched(ct, Lt, n_micro=nm, max_steps=20, seed=11, stop_early=False)
        assert torch.equal(seq_a, b["state"]["seq"]) and torch.equal(len_a, b["state"]["len"])
        assert torch.equal(lp_a, b["state"]["logp"])


def test_chain_streams_are_distinct_and_probe_is_bounded(dev):
    """gpt._chain_streams hands out distinct HIP streams, also when other streams have bound the hardware queues first, keeps a
    bounded probe log, and re-enables the turnstile after a successful re-probe.  (The wall-clock check that the chosen streams
    really overlap lives in tests/test_perf_gpu.py, marker `perf`: timing assertions do not belong in the parity suite.)"""
    from shapeformer_amd import _lib as L
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    decoys = [torch.cuda.Stream(device=dev) for _ in range(5)]      # whatever the process used before
    for s in decoys:
        L.check(L.lib().sfmi_stream_spin(1, s.cuda_stream), "spin")
    torch.cuda.synchronize()
    for _ in range(40):
        S = g._chain_streams(3)
        assert len(S) == 3 and len({s.cuda_stream for s in S}) == 3
    assert len(g._chain_probe) <= 32
    g._mb_shared_queue = True                 # as if a probe had failed: every 64th use re-probes from scratch
    for _ in range(64):
        g._chain_streams(3)
    assert len(g._mb_streams) >= 3
    assert L.lib().sfmi_stream_spin(-1, None) == -1            # SFMI_EINVAL
    gen = L.lib().sfmi_tune_generation()
    assert L.lib().sfmi_tune_set(b"dgemm_un", 0) == 0 and L.lib().sfmi_tune_generation() == gen + 1
    assert L.lib().sfmi_tune_set(b"dgemm_nw", 5) == -1 and L.lib().sfmi_tune_generation() == gen + 1


def test_sample_next_tuple_generator_protocol(dev):
    """mingpt.py:297-310 protocol: next(gen) -> position logits, gen.send(target_pos) -> value logits; must equal the
    teacher-forced forward on the same inputs."""
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    t = np.load(os.path.join(G, "gpt_tiny.npz"))
    c, z, ex = (torch.from_numpy(t[k]) for k in ("c_idx", "z_idx", "extra"))
    cz = torch.cat([c, z], 1)
    L_c = c.shape[1]
    idx, tgt = cz[:, :-1], cz[:, 1:]
    want = g.forward(idx, ex[:, :-1], L_c, tgt)
    w0, w1 = want[0].clone(), want[1].clone()
    gen = g.sample_next_tuple(idx, extra_idx=ex[:, :-1], L_cond=L_c)
    l0 = next(gen)
    assert torch.equal(l0, w0)
    l1 = gen.send(tgt[..., 0])
    assert torch.equal(l1, w1)


# 72 / 90 rows: one chain of 5 / 6 row tiles of the decode GEMM; 105 / 138 / 210 rows: more than a decode launch holds, `sample`
# runs them as 2 / 2 / 3 interleaved chains (the two-n-tile "wide" kernel that used to take them in one launch was retired in
# round 3: it never beat the narrow chains)
@pytest.mark.parametrize("reps", [24, 30, 35, 46, 70])
def test_many_rows_in_one_sample_call_match_the_small_batch(dev, reps):
    """`sample` on 72..210 rows (one 5/6-tile chain, or several interleaved chains behind the same call, teacher forcing and
    logit history included) against the same rows in a 3-row batch: step logits within 2e-4 (bit-identical in practice: a
    row's arithmetic does not depend on the launch shape), forced tokens returned, greedy row equal."""
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    c3, Lc3 = _cond_rows()
    ct3, Lt3 = torch.from_numpy(c3).to(dev, torch.int32), torch.from_numpy(Lc3).to(dev)
    steps = 12
    a = g.sample(ct3, Lt3, max_steps=steps, seed=5, stop_early=False, return_logits=True)
    forced = a["samples"][:, :steps].numpy()
    a = g.sample(ct3, Lt3, max_steps=steps, seed=5, stop_early=False, return_logits=True, force_tokens=forced)
    c = np.concatenate([c3] * reps, 0)
    Lc = np.concatenate([Lc3] * reps, 0)
    ct, Lt = torch.from_numpy(c).to(dev, torch.int32), torch.from_numpy(Lc).to(dev)
    b = g.sample(ct, Lt, max_steps=steps, seed=5, stop_early=False, return_logits=True,
                 force_tokens=np.concatenate([forced] * reps, 0))
    assert b["samples"].shape[0] == 3 * reps and (b["samples"].numpy() == np.concatenate([forced] * reps, 0)).all()
    for s in range(2):
        la, lb = a["logits_history"][s], b["logits_history"][s]
        ref = torch.cat([la] * reps, 0)
        ok = torch.isfinite(ref)
        assert torch.equal(ok, torch.isfinite(lb))
        assert float((lb[ok] - ref[ok]).abs().max()) < 2e-4
    # and the free-running (hipGraph) form completes with the stop rule intact
    d = g.sample(ct, Lt, max_steps=steps, seed=5, stop_early=False)
    assert d["samples"].shape == (3 * reps, steps, 2)
    assert (d["samples"][0] == a["samples"][0]).all()   # greedy row 0 (best_in_first) is batch-size independent here


def test_shared_prefix_sampling_is_bit_identical_to_expanded_rows(dev):
    """sample_n copies of ONE condition (VisShapeFormer.compute_batch, shapeformer.py:222-260): prefill once, the condition's
    keys / values live once (row 0's cache) and every row's decode attention reads them from there - tokens, log-probs and
    masked-logit history must equal the run that prefills and stores the condition S times."""
    from shapeformer_amd.gpt import CondTupleGPT
    sd, sd_t, cfg = _tiny()
    g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    g.PREFILL_BLAS_ROWS = None            # same prefill kernel for 1 and S rows (the library GEMM picks kernels by size)
    c3, Lc3 = _cond_rows()
    for S in (5, 16, 40):
        c = torch.from_numpy(c3[:1, :Lc3[0]]).expand(S, -1, -1).contiguous()
        Lc = torch.full((S,), int(Lc3[0]), dtype=torch.int32)
        a = g.sample(c, Lc, max_steps=30, seed=4, stop_early=False, return_logits=True)
        b = g.sample(c, Lc, max_steps=30, seed=4, stop_early=False, return_logits=True, shared_prefix=True)
        assert torch.equal(a["samples"], b["samples"]) and torch.equal(a["log_prob"], b["log_prob"])
        assert all(torch.equal(x, y) for x, y in zip(a["logits_history"], b["logits_history"]))
        assert len(set(map(tuple, a["samples"][1:, :, 0].tolist()))) > 1          # the stochastic rows do differ from each other
        # "auto" (what the sample_n callers pass): the shared form from SHARED_PREFIX_MIN_ROW_TOKENS rows x condition tokens on
        taken = []
        orig = g.prefill
        g.prefill = lambda st, rows, P, _o=orig: (taken.append(rows), _o(st, rows, P))[1]
        try:
            for thr in (1, 10 ** 9):
                g.SHARED_PREFIX_MIN_ROW_TOKENS = thr
                d = g.sample(c, Lc, max_steps=30, seed=4, stop_early=False, shared_prefix="auto")
                assert torch.equal(a["samples"], d["samples"]) and torch.equal(a["log_prob"], d["log_prob"])
        finally:
            g.prefill = orig
            del g.SHARED_PREFIX_MIN_ROW_TOKENS
        assert taken[0] == 1 and taken[1] == S, taken      # one prefilled row when shared, S when expanded
    with pytest.raises(AssertionError):
        g.sample(torch.from_numpy(c3), torch.from_numpy(Lc3), max_steps=4, shared_prefix=True)   # different rows: refused


def test_attention_turnstile_is_scheduling_only(dev):
    """ATTN_LANES (csrc/gpt.hip:attn_gate_kernel): three interleaved chains behind a one-lane turnstile give exactly the
    tokens / log-probs of the ungated run, every gate was passed in order (tickets == launches) and none timed out."""
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=400)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=400, device=dev)
    rs = np.random.RandomState(5)
    B, steps = 36, 12
    Lc = rs.randint(8, 30, B).astype(np.int32)
    tok = np.full((B, 64, 2), 4096, np.int32)
    for b in range(B):
        tok[b, :Lc[b] - 1, 0] = np.sort(rs.choice(4096, Lc[b] - 1, replace=False)); tok[b, :Lc[b] - 1, 1] = rs.randint(0, 4096, Lc[b] - 1)
    ct, lt = torch.from_numpy(tok), torch.from_numpy(Lc)
    ref = g.sample_microbatched(ct, lt, n_micro=3, max_steps=steps, stop_early=False, seed=3)
    want = {k: v.clone() for k, v in ref["state"].items()}
    g.ATTN_LANES = 1
    try:
        got = g.sample_microbatched(ct, lt, n_micro=3, max_steps=steps, stop_early=False, seed=3)
    finally:
        g.ATTN_LANES = 0
    for k in ("seq", "len", "logp"):
        assert torch.equal(got["state"][k], want[k]), k
    sem = g._sem.cpu().tolist()
    assert sem[0] == sem[1] == 3 * 3 * steps and sem[2] == 0, sem      # 3 chains x 3 layers x steps launches, no time-outs


def test_more_rows_than_the_chains_hold_run_as_rounds(dev):
    """400 rows on 2 chains (2 x 192 rows at most) = two successive rounds of 2 x 100-row chains (each two row groups of the decode
    GEMM); the same rows as 3 chains in one round must give the same tokens / log-probs (uniforms and the greedy row are indexed
    by GLOBAL row in every round)."""
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=400)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=400, device=dev)
    rs = np.random.RandomState(8)
    B, steps = 400, 6
    Lc = rs.randint(6, 20, B).astype(np.int32)
    tok = np.full((B, 32, 2), 4096, np.int32)
    for b in range(B):
        tok[b, :Lc[b] - 1, 0] = np.sort(rs.choice(4096, Lc[b] - 1, replace=False)); tok[b, :Lc[b] - 1, 1] = rs.randint(0, 4096, Lc[b] - 1)
    ct, lt = torch.from_numpy(tok), torch.from_numpy(Lc)
    a = g.sample_microbatched(ct, lt, n_micro=3, max_steps=steps, stop_early=False, seed=4)
    want = {k: v.clone() for k, v in a["state"].items()}
    b2 = g.sample_microbatched(ct, lt, n_micro=2, max_steps=steps, stop_early=False, seed=4)
    assert b2["steps"] == steps and b2["state"]["seq"].shape[0] == B
    for k in ("seq", "len", "Lc", "logp"):
        assert torch.equal(b2["state"][k], want[k]), k
    # and through `sample` (host result), which splits the rows into 4 chains
    h = g.sample(ct, lt, max_steps=steps, stop_early=False, seed=4)
    seq = want["seq"].cpu()
    for r in (0, 57, 133, 399):
        assert torch.equal(h["samples"][r], seq[r, int(Lc[r]):int(Lc[r]) + steps].long())


def test_rounds_that_stop_early_are_padded_like_a_single_run(dev):
    """Early exit (shapeformer.py:110-115: the loop stops once EVERY row has drawn an end token) with more rows than one round of
    chains holds: the rounds stop independently, a single run would have kept stepping the rows that ended first - which can only
    draw (end, end) pairs with log-probability 0 from then on.  The merged result of three rounds must equal the single run's:
    same step count, same tokens (the end-token padding included), same lengths and log-probs."""
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=400)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=400, device=dev)
    rs = np.random.RandomState(18)
    B = 80
    Lc = rs.randint(6, 20, B).astype(np.int32)
    tok = np.full((B, 32, 2), 4096, np.int32)
    for b in range(B):
        tok[b, :Lc[b] - 1, 0] = np.sort(rs.choice(2048, Lc[b] - 1, replace=False)); tok[b, :Lc[b] - 1, 1] = rs.randint(0, 4096, Lc[b] - 1)
    ct, lt = torch.from_numpy(tok), torch.from_numpy(Lc)
    kws = dict(n_micro=4, max_steps=240, stop_early=True, check_every=8, seed=12, mask_invalid_completion=False)
    one = g.sample_microbatched(ct, lt, **kws)                       # 4 chains of 20 rows: one round
    want = {k: v.clone() for k, v in one["state"].items()}
    assert one["steps"] < 240, "the rows must end by themselves for this test to mean anything"
    old = g.MAX_CHAIN_ROWS
    try:
        g.MAX_CHAIN_ROWS = 8                                         # 4 chains x 8 rows = 32 rows per round -> three rounds
        many = g.sample_microbatched(ct, lt, **kws)
    finally:
        g.MAX_CHAIN_ROWS = old
    assert many["steps"] == one["steps"]
    for k in ("len", "Lc", "seq", "logp"):
        assert torch.equal(many["state"][k], want[k]), k
    ln, seq = want["len"].cpu().numpy(), want["seq"].cpu().numpy()
    assert all(tuple(seq[b, ln[b] - 1]) == (4096, 4096) for b in range(B))      # every row ends on the end-token pair
    stops = sorted({int(np.argmax((seq[b, Lc[b]:ln[b], 0] == 4096))) for b in range(B)})
    assert len(stops) > 3, "rows should end at different steps"


def test_large_batches_keep_the_single_chain_features(dev):
    """More than 96 rows in one `sample` call run as interleaved chains; the features of the single-chain path must survive the
    split: shared-prefix KV (every chain prefills the common condition once) and a non-empty z prefix, both bit-identical to the
    plain expanded run of the same rows."""
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=400)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=400, device=dev)
    rs = np.random.RandomState(9)
    S, Lc, steps = 120, 14, 7
    c = np.full((1, Lc, 2), 4096, np.int32)
    c[0, :Lc - 1, 0] = np.sort(rs.choice(4096, Lc - 1, replace=False)); c[0, :Lc - 1, 1] = rs.randint(0, 4096, Lc - 1)
    ct = torch.from_numpy(np.repeat(c, S, 0)); lt = torch.full((S,), Lc, dtype=torch.int32)
    a = g.sample(ct, lt, max_steps=steps, seed=6, stop_early=False, shared_prefix=False)
    b = g.sample(ct, lt, max_steps=steps, seed=6, stop_early=False, shared_prefix=True)
    assert a["samples"].shape == (S, steps, 2) and torch.equal(a["samples"], b["samples"]) and torch.equal(a["log_prob"], b["log_prob"])
    assert len({tuple(r.flatten().tolist()) for r in a["samples"][1:]}) > 1       # the stochastic rows really differ from each other
    z = a["samples"][:, :3].to(torch.int32)                                        # continue every row after its own first 3 tokens
    zc = g.sample(ct, lt, max_steps=steps - 3, seed=8, stop_early=False, z_tokens=z)
    assert zc["samples"].shape == (S, steps, 2) and torch.equal(zc["samples"][:, :3], a["samples"][:, :3])
    one = g.sample(ct[:5], lt[:5], max_steps=steps - 3, seed=8, stop_early=False, z_tokens=z[:5])
    # rows 0..4 as a 5-row batch: row 0 is greedy in both; rows 1..4 draw with uniforms indexed by (global row, rows_total), so only
    # the greedy row is comparable across batch sizes
    assert torch.equal(one["samples"][0], zc["samples"][0])


@pytest.mark.parametrize("M", [48, 80, 96, 130, 192])
def test_decode_gemm_two_n_tile_form_and_row_groups_are_bit_identical(dev, M):
    """csrc/gpt.hip dgemm_kernel<MT,8,2,NT=2> (two n-tiles per wave, row groups of <= 3 row tiles; tuning knob dgemm_nt2) against
    the one-tile form (row groups of <= 6 row tiles above 96 rows) on the decode step's GEMM shapes (LN fold, GELU, residual,
    split-K 4, the odd 257-tile head): same k order and accumulator chains, so every output bit must agree; rows >= M untouched
    in the row-major head output."""
    from shapeformer_amd import _lib as L
    from shapeformer_amd.gpt import pack_skinny16
    lib = L.lib()
    g = torch.Generator(device="cpu").manual_seed(M)
    Mp = int(lib.sfmi_decode_gemm_padded_rows(M))
    assert Mp >= M and Mp % 16 == 0
    try:
        for (N, K, ln, act, use_res, S) in [(3072, 1024, 1, 0, False, 1), (1024, 1024, 0, 0, True, 1), (4096, 1024, 1, 1, False, 1),
                                            (1024, 4096, 0, 0, True, 4), (4097, 1024, 1, 0, False, 1)]:
            Np = (N + 15) // 16 * 16
            wp = pack_skinny16(torch.randn(N, K, generator=g) * 0.05).to(dev)
            x = torch.randn(Mp * K, generator=g).to(dev)                  # fragment-packed activations: any values do
            c1, c2 = torch.randn(Np, generator=g).to(dev), torch.randn(Np, generator=g).to(dev)
            packed = 0 if N == 4097 else 1
            ldo = 4128 if N == 4097 else N
            res = torch.randn(Mp * N, generator=g).to(dev) if use_res else None
            slab = torch.empty(lib.sfmi_decode_gemm_slab_floats(Mp, 4096, 4), device=dev)
            cnt = torch.zeros(Mp // 16 * 260, device=dev, dtype=torch.int32)
            outs = []
            for nt2 in (0, 2):          # 0 = one n-tile per wave, 2 = two n-tiles whenever possible (the default, 1, picks per shape)
                L.check(lib.sfmi_tune_set(b"dgemm_nt2", nt2), "tune")
                out = torch.full((Mp * max(N, ldo),), 7.0, device=dev)
                L.check(lib.sfmi_decode_gemm_f32(L.ptr(x), L.ptr(wp), L.ptr(c1) if ln else None, L.ptr(c2), L.ptr(res), L.ptr(out), M, N, K, ldo,
                                                 ln, act, packed, S, L.ptr(slab) if S > 1 else None, L.ptr(cnt) if S > 1 else None,
                                                 L.stream_ptr()), "sfmi_decode_gemm_f32")
                torch.cuda.synchronize()
                outs.append(out.cpu())
            a, b = outs
            if packed:      # padded row tiles hold garbage of the padded inputs in both forms; compare the tiles that contain real rows
                nt_real = (M + 15) // 16 * (N // 16) * 256
                a, b = a[:nt_real], b[:nt_real]
            else:
                a, b = a[:M * ldo], b[:M * ldo]
                assert bool((outs[1][M * ldo:Mp * ldo] == 7.0).all()), "rows beyond M must not be written in the row-major form"
            assert torch.isfinite(a).all() and float(a.abs().max()) > 0
            assert torch.equal(a, b), (M, N, K, float((a - b).abs().max()))
    finally:
        L.check(lib.sfmi_tune_set(b"dgemm_nt2", 1), "tune")


@pytest.mark.parametrize("N,K", [(3072, 1024), (50, 64), (4097, 1024)])
def test_ln_fold_pack_kernel_against_host_pack_and_float64_sums(dev, N, K):
    """csrc/gpt.hip ln_fold_pack_kernel (the decode weights' LayerNorm fold, mingpt.py:103-111): the fragment-ordered matrix must be the
    host packer's (`sfmi_skinny16_pack_weight`) of W diag(gamma) bit for bit - rows beyond N zero - and c1 / c2 the float64 row sums
    rounded once; gamma = beta = NULL is the plain pack."""
    from shapeformer_amd import _lib as L
    lib = L.lib()
    rs = np.random.RandomState(N + K)
    W = (rs.randn(N, K) * 0.05).astype(np.float32)
    gam, bet, bias = (1 + 0.1 * rs.randn(K)).astype(np.float32), (0.1 * rs.randn(K)).astype(np.float32), rs.randn(N).astype(np.float32)
    Np = (N + 15) // 16 * 16
    t = lambda a: torch.from_numpy(a).to(dev)
    Wd, gd, bd, biasd = t(W), t(gam), t(bet), t(bias)
    wp, c1, c2 = torch.empty(Np * K, device=dev), torch.empty(Np, device=dev), torch.empty(Np, device=dev)
    L.check(lib.sfmi_ln_fold_pack_f32(L.ptr(Wd), L.ptr(gd), L.ptr(bd), L.ptr(biasd), L.ptr(wp), L.ptr(c1), L.ptr(c2), N, K, L.stream_ptr()), "fold")
    Wg = W * gam[None, :]                      # the f32 products the kernel forms
    want = np.empty(Np * K, np.float32)
    L.check(lib.sfmi_skinny16_pack_weight(Wg.ctypes.data, N, K, want.ctypes.data), "pack")
    assert np.array_equal(wp.cpu().numpy(), want)
    r1 = Wg.astype(np.float64).sum(1)
    r2 = (W.astype(np.float64) * bet.astype(np.float64)[None, :]).sum(1) + bias
    assert np.array_equal(c1.cpu().numpy()[N:], np.zeros(Np - N, np.float32)) and np.array_equal(c2.cpu().numpy()[N:], np.zeros(Np - N, np.float32))
    # float64 sums in another association order: equal after the single rounding to f32 up to one ulp
    assert np.allclose(c1.cpu().numpy()[:N], r1.astype(np.float32), rtol=2e-7, atol=1e-9)
    assert np.allclose(c2.cpu().numpy()[:N], r2.astype(np.float32), rtol=2e-7, atol=1e-9)
    wp2 = torch.empty(Np * K, device=dev)
    L.check(lib.sfmi_ln_fold_pack_f32(L.ptr(Wd), None, None, None, L.ptr(wp2), None, None, N, K, L.stream_ptr()), "pack only")
    L.check(lib.sfmi_skinny16_pack_weight(W.ctypes.data, N, K, want.ctypes.data), "pack")
    assert np.array_equal(wp2.cpu().numpy(), want)
