"""Pin the CPU oracle to the REAL reference and (re)generate tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference, imported via oracle/refimport.py
with stub modules — nothing from the reference is copied).  For every stage it
  1. runs the reference's own code on seeded inputs + hash weights,
  2. runs the oracle restatement on the same inputs and asserts agreement
     (exact for integers, fp32 round-off for floats),
  3. stores small input/expected-output vectors as fixtures for the tests that run
     where the reference is absent (CPU CI and the GPU box).

    python -m oracle.make_golden
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import gpt_oracle as GO          # noqa: E402
from oracle import refimport as R            # noqa: E402
from oracle import tokens_oracle as TO       # noqa: E402
from oracle import vqdif_oracle as VO        # noqa: E402
from shapeformer_amd import weights as W     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TINY = dict(n_embd=64, n_head=4, n_layers=[2, 1], block_size=96)


def demo_cloud(name, kind, n):
    a = np.load(os.path.join(R.REF_ROOT, "demo", "dataset", name, kind + ".npy")).astype(np.float32)
    step = max(1, a.shape[0] // n)
    return a[::step][:n]


def section(msg):
    print(f"\n== {msg}", flush=True)


@torch.no_grad()
def vqdif_fixtures():
    section("VQDIF res16: reference vs oracle (encode, quantize, tokens, decode)")
    m = R.build_vqdif(16)
    sd = VO.to_torch_sd(W.make_state_dict(W.vqdif_spec(16)))
    from shapeformer.models import common as RC
    n = 2048
    cloud = np.stack([demo_cloud("car", "Xbd", n), demo_cloud("armchair", "Xbd", n)])
    X = torch.from_numpy(cloud)
    rq, rmode, renc = m.quantize_cloud(X)
    oq, omode, oenc = VO.quantize_cloud(sd, X)
    rf, rmask = m.encode(X)
    assert int(rmode) == int(omode)
    assert torch.equal(rq, oq) and torch.equal(renc["quant_ind"], oenc["quant_ind"])
    assert torch.equal(rmask, oenc["grid_mask"])
    fd = (rf - oenc["grid_feat"]).abs().max().item()
    print(f"quant_ind exact ({int(rmask.sum())} occupied cells), latent max|diff| {fd:.2e} (scale {rf.abs().max():.1f})")
    assert fd < 1e-3
    # per-point encoder stages (oracle only; pinned transitively by the exact indices above)
    c, cell, u, stages = VO.encoder_points(sd, X / 2.0, return_stages=True)

    end = (4096, 4096)
    rs, rmode2 = RC.batch_dense2sparse(rq, max_length=512, end_tokens=torch.tensor(end))
    os_, omode2 = TO.batch_dense2sparse(rq.numpy(), max_length=512, end_tokens=end)
    assert int(rmode2) == omode2 and np.array_equal(rs.numpy(), os_)
    rp = RC.pack_sparse(rs, end_tokens=end)
    op = TO.pack_sparse(os_, end_tokens=end)
    assert np.array_equal(rp.numpy(), op)
    rd = RC.batch_sparse2dense(rp, empty_ind=rmode2, dense_res=16)
    od = TO.batch_sparse2dense(op, omode2, 16, batch_size=2)
    assert np.array_equal(rd.numpy(), od) and np.array_equal(od, rq.numpy())
    rs40, _ = RC.batch_dense2sparse(rq, max_length=40, end_tokens=torch.tensor(end))
    os40, _ = TO.batch_dense2sparse(rq.numpy(), max_length=40, end_tokens=end)
    assert np.array_equal(rs40.numpy(), os40)
    print(f"tokens exact: L={os_.shape[1]}, truncated L=40 exact, dense round trip exact")

    Q = 24
    Xtg = torch.from_numpy(VO.make_grid(Q))[None].expand(2, -1, -1)
    rl = m.decode_index(rq, Xtg)["logits"]
    ol = VO.decode_index(sd, oq, Xtg)
    ld = (rl - ol).abs().max().item()
    print(f"decode_index logits max|diff| {ld:.2e} (scale {rl.abs().max():.1f})")
    assert ld < 1e-4
    grid = VO.decoder_grid(sd, VO.get_code(sd, oq))
    rgrid = m.decoder.upsampler(m.decoder.unet3d(m.quantizer.get_code(rq)))
    gd = (grid - rgrid).abs().max().item()
    assert gd < 1e-4
    sel = np.arange(0, 64 ** 3, 997)
    np.savez_compressed(
        os.path.join(OUT, "vqdif16_small.npz"),
        cloud=cloud, quant_ind=rq.numpy().astype(np.int16), quant_ind_raw=renc["quant_ind"].numpy().astype(np.int16),
        mode=np.int64(int(rmode)), grid_mask=np.packbits(rmask.numpy()), cell=cell.numpy().astype(np.int32),
        latent_sel=rf.permute(0, 2, 3, 4, 1).reshape(2, -1, 128)[:, ::61].numpy(),
        latent_abs_sum=rf.abs().sum(dim=(1, 2, 3, 4)).numpy(),
        enc_stage1_sel=stages[1][:, ::64].numpy(), enc_stage4_sel=stages[4][:, ::64].numpy(),
        enc_c_sel=c[:, ::64].numpy(),
        tokens=rs.numpy().astype(np.int32), tokens_L40=rs40.numpy().astype(np.int32), packed=rp.numpy().astype(np.int32),
        mode2=np.int64(int(rmode2)),
        dec_grid_sel=rgrid.permute(0, 2, 3, 4, 1).reshape(2, -1, 32)[:, sel].numpy(), dec_grid_sel_idx=sel,
        dec_grid_abs_sum=rgrid.abs().sum(dim=(1, 2, 3, 4)).numpy(),
        Q=np.int64(Q), logits=rl.numpy()[..., 0])
    return m, rq, rs


@torch.no_grad()
def vqdif32_fixture():
    section("VQDIF res32: reference vs oracle")
    m = R.build_vqdif(32)
    sd = VO.to_torch_sd(W.make_state_dict(W.vqdif_spec(32)))
    cloud = demo_cloud("sofa", "Xbd", 2048)[None]
    X = torch.from_numpy(cloud)
    rq, rmode, renc = m.quantize_cloud(X)
    oq, omode, oenc = VO.quantize_cloud(sd, X)
    assert torch.equal(rq, oq) and int(rmode) == int(omode)
    Q = 16
    Xtg = torch.from_numpy(VO.make_grid(Q))[None]
    rl = m.decode_index(rq, Xtg)["logits"]
    ol = VO.decode_index(sd, oq, Xtg)
    assert (rl - ol).abs().max().item() < 1e-4
    print(f"res32 quant_ind exact ({int(renc['grid_mask'].sum())} cells), logits diff {(rl - ol).abs().max():.1e}")
    np.savez_compressed(os.path.join(OUT, "vqdif32_small.npz"), cloud=cloud,
                        quant_ind=rq.numpy().astype(np.int16), mode=np.int64(int(rmode)),
                        grid_mask=np.packbits(renc["grid_mask"].numpy()), Q=np.int64(Q), logits=rl.numpy()[..., 0])


def token_known_answers():
    section("token packing / sampling filter: reference known-answer cases (SURVEY §4)")
    R.setup()
    from shapeformer.models import common as RC
    RC.pack_unpack_unittest()
    sp = np.array([[0, 4, 1], [0, 5, 2], [1, 1, 5], [2, 3, 2], [2, 5, 1], [3, 0, 0]])
    ru = RC.unpack_sparse(torch.from_numpy(sp)).numpy()
    ou = TO.unpack_sparse(sp)
    assert np.array_equal(ru, ou)
    testA = np.ones((2, 2, 2, 2), np.int64)
    testA[0, 1, 1, 1], testA[0, 1, 1, 0], testA[0, 1, 0, 0], testA[1, 0, 0, 0], testA[1, 0, 0, 1] = 2, 3, 4, 7, 2
    rp, rm = RC.batch_dense2sparse(torch.from_numpy(testA), unpack=False)
    op, om = TO.dense2packed(testA)
    assert np.array_equal(rp.numpy(), op) and int(rm) == om
    cases = [([1.01, 1, 1.02], 3, .5, 1.), ([2, 1, 0, -1], 3, .7, 1.), ([2, 1, 0, -1], 2, .99, .5)]
    g = np.random.RandomState(0)
    big = g.randn(6, 4097).astype(np.float32) * 3
    filt_in, filt_out, filt_par = [], [], []
    for lg, k, p, t in cases + [(big[i], [100, 100, 300, 1, 100, 50][i], [.4, .9, .9, .001, .05, .4][i],
                                 [1., 1., .7, 1., 1., 2.][i]) for i in range(6)]:
        a = np.asarray(lg, np.float32)
        r = RC.filter_sampling_logits(torch.from_numpy(a.copy()), top_k=k, top_p=p, temperature=t).numpy()
        o = TO.filter_sampling_logits(a, k, p, t)
        assert np.array_equal(r, o), (r, o)
        if a.size > 10:
            filt_in.append(a); filt_out.append(r); filt_par.append((k, p, t))
    # get_next_cond / AR_N extra indices
    from shapeformer.models.shapeformer.representers import get_next_cond
    c_pos = np.array([[3, 9, 20, 4096], [0, 1, 2, 4096]])
    z_pos = np.array([[0, 3, 4, 19, 20, 21, 4096], [5, 6, 7, 8, 9, 4096, 4096]])
    rn = get_next_cond(torch.from_numpy(c_pos), torch.from_numpy(z_pos), 4096).numpy()
    on = TO.get_next_cond(c_pos, z_pos, 4096)
    assert np.array_equal(rn, on)
    print("all known-answer cases exact")
    np.savez_compressed(os.path.join(OUT, "tokens_known.npz"), sp=sp, unpacked=ru, testA=testA, packedA=rp.numpy(),
                        modeA=np.int64(int(rm)), filt_in=np.stack(filt_in), filt_out=np.stack(filt_out),
                        filt_par=np.array(filt_par, np.float64), c_pos=c_pos, z_pos=z_pos, next_cond=rn)


@torch.no_grad()
def gpt_fixtures(vq, rq, tokens):
    section("CondTupleGPT tiny config: reference forward + reference sample_indices vs oracle")
    sf = R.build_shapeformer(vq=vq, **TINY)
    gsd = VO.to_torch_sd(W.make_state_dict(W.gpt_spec(n_embd=64, n_layers=(2, 1), block_size=96)))
    cfg = GO.GPTCfg(n_embd=64, n_head=4, n_layers=(2, 1), block_size=96)
    # tokens: truncated condition (L=24) from the car/armchair clouds, target = next tokens
    from shapeformer.models import common as RC
    c_idx, _ = RC.batch_dense2sparse(rq, max_length=24, end_tokens=torch.tensor((4096, 4096)))
    z_idx, _ = RC.batch_dense2sparse(rq, max_length=40, end_tokens=torch.tensor((4096, 4096)))
    extra = sf.representer.get_extra_indices(c_idx, z_idx)
    oextra = TO.extra_indices_AR_N(c_idx.numpy(), z_idx.numpy(), 4096)
    assert np.array_equal(extra.numpy(), oextra)
    cz = torch.cat([c_idx, z_idx], 1)
    L_c = c_idx.shape[1]
    rlog = sf.transformer(idx=cz[:, :-1], extra_idx=extra[:, :-1], L_cond=L_c, target_idx=cz[:, 1:])
    olog = GO.forward_logits(gsd, cfg, cz[:, :-1], extra[:, :-1], L_c, cz[:, 1:])
    d = max((a - b).abs().max().item() for a, b in zip(rlog, olog))
    print(f"teacher-forced logits max|diff| {d:.2e} (scale {rlog[0].abs().max():.1f})")
    assert d < 1e-4
    # loss
    sf.train(False)
    rl, rt = sf(stage="test", Xct=None) if False else (None, None)
    oloss = GO.training_loss(gsd, cfg, c_idx, z_idx, extra).item()
    logits_cut = [l[..., L_c - 1:, :] for l in rlog]
    rloss = sum(torch.nn.functional.cross_entropy(l.reshape(-1, l.shape[-1]), z_idx[..., i].reshape(-1))
                for i, l in enumerate(logits_cut)).item() / 2
    assert abs(oloss - rloss) < 1e-5 * max(1, abs(rloss)), (oloss, rloss)

    # sampling: reference loop (stochastic rows use torch RNG -> only greedy row 0 is comparable)
    S, steps = 3, 20
    c1 = c_idx[:1].expand(S, -1, -1).contiguous()
    torch.manual_seed(0)
    rx, rhist = sf.sample_indices(c_indices=c1, z_indices=c1[:, :0], max_steps=steps, best_in_first=True,
                                  top_k=100, top_p=0.4, temperature=1.0, mask_invalid=True,
                                  mask_invalid_completion=True)
    u = GO.uniforms(0, steps, S)
    for use_cache in (False, True):
        # feed the reference's sampled tokens for rows 1.. (teacher-forced) so every row's logits compare
        ox, ohist, _ = GO.sample_indices(gsd, cfg, c1, steps, u, use_cache=use_cache, force_tokens=rx.numpy(),
                                         stop_early=False)
        n = rx.shape[1]
        for i in range(2):
            a, b = rhist[i].numpy(), ohist[i][:, :n]
            fin = np.isfinite(a)
            assert np.array_equal(fin, np.isfinite(b))
            dd = np.abs(a[fin] - b[fin]).max()
            assert dd < 2e-4, dd
        print(f"sampling logits_history (cache={use_cache}) max|diff| {dd:.2e}; masks identical")
    og, oh, _ = GO.sample_indices(gsd, cfg, c1[:1], steps, u[:, :, :1], use_cache=True, stop_early=False)
    assert np.array_equal(og[0, :rx.shape[1]], rx[0].numpy()), "greedy row diverged"
    print(f"greedy row: {rx.shape[1]} steps token-exact vs reference")
    # stochastic rows under injected uniforms (oracle-defined inverse CDF; kernel is checked against this)
    os_, oshist, _ = GO.sample_indices(gsd, cfg, c1, steps, u, use_cache=True, stop_early=False)
    lp = GO.compute_log_probs(os_, oshist)
    from shapeformer.models.shapeformer.shapeformer import compute_log_probs as r_clp
    rlp = r_clp(os_, oshist)
    assert np.allclose(lp, rlp, atol=1e-5, equal_nan=True)
    np.savez_compressed(os.path.join(OUT, "gpt_tiny.npz"), c_idx=c_idx.numpy(), z_idx=z_idx.numpy(),
                        extra=extra.numpy(), logits0_sel=rlog[0][:, ::7, ::41].numpy(),
                        logits1_sel=rlog[1][:, ::7, ::41].numpy(), loss=np.float64(rloss),
                        ref_sampled=rx.numpy(), ref_hist0_row0=rhist[0][0].numpy(), ref_hist1_row0=rhist[1][0].numpy(),
                        steps=np.int64(steps), orc_sampled=os_, orc_logprob=lp)

    # continuing a NON-EMPTY z_indices (shapeformer.py:60-70): the reference copies cat(c, z) into `sampled` and generates after it;
    # its step counter (masker's step_j) restarts at 0, so the first new position is not constrained by mask_invalid
    Lz, steps2 = 5, 10
    zp = z_idx[:1, :Lz].expand(S, -1, -1).contiguous()
    torch.manual_seed(1)
    rx2, rhist2 = sf.sample_indices(c_indices=c1, z_indices=zp, max_steps=steps2, best_in_first=True, top_k=100, top_p=0.4,
                                    temperature=1.0, mask_invalid=True, mask_invalid_completion=True)
    assert np.array_equal(rx2[:, :Lz].numpy(), zp.numpy()), "the returned tokens start with the z prefix"
    n2 = rx2.shape[1] - Lz
    u2 = GO.uniforms(1, steps2, S)
    for use_cache in (False, True):
        ox2, ohist2, _ = GO.sample_indices(gsd, cfg, c1, steps2, u2, use_cache=use_cache, force_tokens=rx2[:, Lz:].numpy(),
                                           stop_early=False, z_indices=zp)
        assert np.array_equal(ox2[:, :Lz + n2], rx2.numpy())
        for i in range(2):
            a, b = rhist2[i].numpy(), ohist2[i][:, :n2]
            fin = np.isfinite(a)
            assert np.array_equal(fin, np.isfinite(b)), "z-prefix continuation: masks differ"
            dd = np.abs(a[fin] - b[fin]).max()
            assert dd < 2e-4, dd
        print(f"z-prefix continuation logits_history (cache={use_cache}) max|diff| {dd:.2e}; masks identical")
    og2, _, _ = GO.sample_indices(gsd, cfg, c1[:1], steps2, u2[:, :, :1], use_cache=True, stop_early=False, z_indices=zp[:1])
    assert np.array_equal(og2[0, :rx2.shape[1]], rx2[0].numpy()), "greedy row diverged after a z prefix"
    print(f"greedy row after a {Lz}-token z prefix: {n2} steps token-exact vs reference")
    os2, oshist2, _ = GO.sample_indices(gsd, cfg, c1, steps2, u2, use_cache=True, stop_early=False, z_indices=zp)
    np.savez_compressed(os.path.join(OUT, "gpt_tiny_zprefix.npz"), c_idx=c1.numpy(), z_prefix=zp.numpy(), ref_sampled=rx2.numpy(),
                        ref_hist0_row0=rhist2[0][0].numpy(), ref_hist1_row0=rhist2[1][0].numpy(), steps=np.int64(steps2),
                        orc_sampled=os2)

    section("CondTupleGPT full config (20+4 layers, d=1024): reference forward vs oracle")
    g = R.build_gpt()
    fsd = VO.to_torch_sd(W.make_state_dict(W.gpt_spec()))
    fcfg = GO.GPTCfg()
    cz2 = cz[:1, :48]
    ex2 = extra[:1, :48]
    rl2 = g(idx=cz2[:, :-1], extra_idx=ex2[:, :-1], L_cond=L_c, target_idx=cz2[:, 1:])
    ol2 = GO.forward_logits(fsd, fcfg, cz2[:, :-1], ex2[:, :-1], L_c, cz2[:, 1:])
    d2 = max((a - b).abs().max().item() for a, b in zip(rl2, ol2))
    print(f"full-size logits max|diff| {d2:.2e} (scale {rl2[0].abs().max():.1f})")
    assert d2 < 1e-3
    pos_sel = np.array([0, 11, 23, 30, 46])
    np.savez_compressed(os.path.join(OUT, "gpt_full_probe.npz"), cz=cz2.numpy(), extra=ex2.numpy(), L_c=np.int64(L_c),
                        pos_sel=pos_sel, logits0=rl2[0][0, pos_sel].numpy(), logits1=rl2[1][0, pos_sel].numpy())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    token_known_answers()
    vq, rq, tokens = vqdif_fixtures()
    vqdif32_fixture()
    gpt_fixtures(vq, rq, tokens)
    print("\nALL ORACLE PINS PASSED; fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
