"""GPU: the reference's YAML/plugin surface end to end — instantiate_from_opt on a config with the shipped YAML's keys
(shapenet_scale.yaml layout, transformer shrunk for test speed), completion and one training step."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _opt():
    P = "shapeformer.models.shapeformer."
    return {"expr_name": "shapeformer/test", "pl_model_opt": {"class": P + "shapeformer.ShapeFormer", "kwargs": dict(
        voxel_res=16, end_tokens=[4096, 4096], vocab_sizes=[4097, 4097], extra_vocab_sizes=[4097], block_size=500, tuple_n=2,
        representer_opt={"class": P + "representers.AR_N", "kwargs": dict(
            voxel_res=16, uncond=False, no_val_ind=False, block_size=500, end_tokens=[4096, 4096], random_cind_masking=True,
            mask_invalid_completion=True, allow_generated_weights=True,   # no checkpoints ship: opt in to hash weights
            vqvae_opt={"class": "shapeformer.models.vqdif.vqdif.VQDIF", "ckpt_path": "experiments/none.ckpt",
                       "yaml_path": "configs/vqdif/shapenet_res16.yaml"})},
        transformer_opt={"class": P + "transformer.mingpt.CondTupleGPT", "kwargs": dict(
            tuple_n=2, vocab_sizes=[4097, 4097], extra_vocab_sizes=[4097], n_layers=[2, 1], block_size=500, n_head=2, n_embd=128,
            attn_pdrop=.01, resid_pdrop=.01, embd_pdrop=.01)},
        optim_opt=dict(lr=1e-3, scheduler="StepLR", step_size=10, gamma=.9))}}


def test_instantiate_complete_and_train(dev):
    from shapeformer_amd import plugin as P, synthetic
    opt = P.get_opt(_opt())
    model = P.instantiate_from_opt(opt["pl_model_opt"])
    assert isinstance(model, P.ShapeFormerModel) and model.block_size == 500
    b = synthetic.make_batch(5, 2, n_full=8192, n_partial=4096)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    out = model.complete(batch["Xct"], max_steps=8, decode_res=32, stop_early=False)
    assert out["occupancy"].shape == (2, 32 ** 3) and bool(torch.isfinite(out["occupancy"]).all())
    assert float(out["occupancy"].min()) >= 0.0 and float(out["occupancy"].max()) <= 1.0
    np.random.seed(0)
    model.make_trainer(opt["pl_model_opt"]["kwargs"]["optim_opt"])
    l = [model.training_step(batch).item() for _ in range(4)]
    assert all(np.isfinite(l)) and l[-1] < l[0]


class _Items:
    """Dataset stand-in: dict items of numpy arrays like the reference datasets' __getitem__."""

    def __init__(self, n):
        from shapeformer_amd import synthetic
        self.items = [synthetic.make_shape(40 + i, n_full=8192, n_partial=4096) for i in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_inference_callbacks_compute_cache_and_export(dev, tmp_path):
    """VisShapeFormer / VisSparseRecon3D resolved from their reference dotted names: compute_batch dict
    keys, computed/<name>.npy cache (reloaded with load_compute), meshes/*.ply and eval/*.npz, probability order."""
    from shapeformer_amd import plugin as P, meshio, callbacks as CB
    opt = P.get_opt(_opt())
    model = P.instantiate_from_opt(opt["pl_model_opt"])
    data = _Items(3)
    cb = P.instantiate_from_opt({"class": "shapeformer.models.shapeformer.shapeformer.VisShapeFormer", "kwargs": dict(
        no_sanity_check=True, every_n_epoch=4, end_tokens=[4096, 4096], top_k=100, top_p=0.4, depth=4, resolution=[256, 256],
        render_samples=32, visual_indices=[0, 2], sample_n=4, sample_max_step=12, decode_res=32, data_dir=str(tmp_path / "sf"),
        keep_logits_history=True)})
    out = cb.process(model, data)
    assert sorted(out) == ["0", "2"]
    comp = np.load(tmp_path / "sf" / "computed" / "0.npy", allow_pickle=True).item()
    assert set(comp) >= {"batch", "samples", "origin_samples", "logits_history", "c_ind", "z_ind", "empty_index", "log_prob"}
    S, Ls, _ = comp["samples"].shape
    assert S == 4 and comp["log_prob"].shape == (4, Ls, 2) and comp["c_ind"].shape[0] == 1
    assert (comp["c_ind"][0, -1] == 4096).all()                                  # condition ends with the end-token pair
    # the device-side log-probabilities are the reference's compute_log_probs of the recorded step logits
    want = CB.compute_log_probs(comp["samples"], comp["logits_history"])
    live = np.isfinite(want)
    assert np.abs(comp["log_prob"][live] - want[live]).max() < 1e-4
    order = cb.sample_order(comp)
    sums = comp["log_prob"].sum((1, 2))
    assert (np.diff(sums[order]) <= 0).all()
    # exported files: one ply per non-degenerate token set, eval npz with the most probable sample first
    plys = sorted(os.listdir(tmp_path / "sf" / "meshes"))
    assert any(p.startswith("0_data_c") for p in plys) and all(p.endswith("_mesh.ply") for p in plys)
    sample_plys = [p for p in plys if p.startswith("0_s")]
    if sample_plys:
        ev = np.load(tmp_path / "sf" / "eval" / "0.npz")
        assert ev["eval_pc"].shape == (100000, 3) and np.array_equal(ev["eval_pc"], ev["recon_0"])
        v, f = meshio.read_ply(str(tmp_path / "sf" / "meshes" / sample_plys[0]))
        assert f.max() < len(v) and np.abs(v).max() <= 1.0
    # load_compute=True reuses the cache: poison the model to prove no recomputation happens
    cb.load_compute = True
    cb.compute_batch = None
    out2 = cb.process(model, data, visual_indices=[0])
    assert sorted(out2["0"]) == sorted(out["0"])
    # VQDIF reconstruction callback
    rc = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VisSparseRecon3D", "kwargs": dict(
        quant_grid_depth=4, decoder_resolution=32, max_length=512, end_tokens=[4096, 4096], visual_indices=[1],
        data_dir=str(tmp_path / "vq"))})
    r = rc.process(model.representer.vqvae_model, data)
    comp = np.load(tmp_path / "vq" / "computed" / "1.npy", allow_pickle=True).item()
    assert set(comp) == {"logits", "quant_ind", "sparse", "grid_mask", "batch"}
    assert comp["logits"].shape == (1, 32 ** 3, 1) and comp["quant_ind"].shape == (1, 16, 16, 16) and comp["sparse"].shape[1] == 3
    assert comp["grid_mask"].dtype == bool and os.path.exists(tmp_path / "vq" / "meshes" / "1.ply")
    assert "recon_mesh" in r["1"]


def test_vqdif_model_training_step_through_the_plugin(dev):
    """vqdif.py:100-105 on the plugin surface: VQDIF resolved from its dotted name, training_step(batch) lowers the loss
    and the inference kernels pick the trained weights up."""
    from shapeformer_amd import plugin as P
    kw = P.default_vqdif_kwargs(16)
    vq = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VQDIF", "kwargs": kw})
    rs = np.random.RandomState(0)
    u = rs.randn(2, 2048, 3)
    Xbd = (u / np.linalg.norm(u, axis=-1, keepdims=True) * 0.5).astype(np.float32)
    Xtg = rs.uniform(-1, 1, (2, 1024, 3)).astype(np.float32)
    Ytg = (np.linalg.norm(Xtg, axis=-1, keepdims=True) < 0.5).astype(np.float32)
    batch = dict(Xbd=Xbd, Xtg=Xtg, Ytg=Ytg)
    vq.make_trainer(dict(lr=1e-3))
    before = vq.decode_index(vq.quantize_cloud(torch.from_numpy(Xbd[:1]).to(dev))[0], Xtg=torch.from_numpy(Xtg[:1]).to(dev))["logits"].clone()
    losses = [float(vq.training_step(batch)) for _ in range(5)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    vq.sync_inference_weights()
    after = vq.decode_index(vq.quantize_cloud(torch.from_numpy(Xbd[:1]).to(dev))[0], Xtg=torch.from_numpy(Xtg[:1]).to(dev))["logits"]
    assert float((after - before).abs().max()) > 1e-4


def test_res32_reconstruction_callback(dev, tmp_path):
    """VQDIF-32 (configs/vqdif/shapenet_res32.yaml layout: one down/up-sampling step, d = 64, 32^3 latent) through the
    reconstruction callback: 32^3 code grid, tokens with the res-32 end tokens, mesh on disk."""
    from shapeformer_amd import plugin as P
    vq = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VQDIF", "kwargs": P.default_vqdif_kwargs(32)})
    rc = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VisSparseRecon3D", "kwargs": dict(
        quant_grid_depth=5, decoder_resolution=48, max_length=2048, end_tokens=[32768, 4096], visual_indices=[0],
        data_dir=str(tmp_path / "vq32"))})
    out = rc.process(vq, _Items(1))
    comp = np.load(tmp_path / "vq32" / "computed" / "0.npy", allow_pickle=True).item()
    assert comp["quant_ind"].shape == (1, 32, 32, 32) and comp["logits"].shape == (1, 48 ** 3, 1)
    assert comp["sparse"][:, 1].max() < 32768 and comp["sparse"][:, 2].max() < 4096
    assert os.path.exists(tmp_path / "vq32" / "meshes" / "0.ply") and "recon_mesh" in out["0"]


def test_checkpoint_resume_is_bit_exact_for_both_models(dev, tmp_path):
    """SURVEY §5 checkpoint/resume: Lightning-layout .ckpt (state_dict with the reference key names + optimizer state);
    a run resumed from the file continues with exactly the losses of the uninterrupted run."""
    from shapeformer_amd import plugin as P, synthetic
    b = synthetic.make_batch(9, 2, n_full=8192, n_partial=4096)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}

    def steps(model, n, seed0):
        out = []
        for i in range(n):
            np.random.seed(seed0 + i)             # get_indices('train') draws the condition subset from numpy
            out.append(float(model.training_step(batch)))
        return out
    opt = P.get_opt(_opt())
    m1 = P.instantiate_from_opt(opt["pl_model_opt"])
    m1.make_trainer(dict(lr=1e-3))
    steps(m1, 2, 0)
    path = m1.save_checkpoint(str(tmp_path / "ck" / "sf.ckpt"), hyper_parameters=opt["pl_model_opt"]["kwargs"])
    want = steps(m1, 2, 2)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert "optimizer_states" not in ck and "sfmi_optimizer_state" in ck and ck["hyper_parameters"]["block_size"] == 500
    assert "transformer.blocks.0.1.attn.key.weight" in ck["state_dict"] and "transformer.blocks.1.0.attn.mask" in ck["state_dict"]
    assert "representer.vqvae_model.encoder.fc_pos.weight" in ck["state_dict"] and ck["global_step"] == 2
    m2 = P.instantiate_from_opt(opt["pl_model_opt"])
    m2.make_trainer(dict(lr=1e-3))
    m2.load_checkpoint(path)
    assert m2.trainer.step_count == 2 and m2.trainer.lr == 1e-3
    assert steps(m2, 2, 2) == want
    # VQDIF: weights + EMA codebook + Adam state
    rs = np.random.RandomState(0)
    Xtg = rs.uniform(-1, 1, (2, 512, 3)).astype(np.float32)
    vb = dict(Xbd=b["Xbd"], Xtg=Xtg, Ytg=(np.linalg.norm(Xtg, axis=-1, keepdims=True) < 0.5).astype(np.float32))
    kw = P.default_vqdif_kwargs(16)
    v1 = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VQDIF", "kwargs": kw})
    v1.make_trainer(dict(lr=1e-3))
    [float(v1.training_step(vb)) for _ in range(2)]
    vpath = v1.save_checkpoint(str(tmp_path / "ck" / "vq.ckpt"))
    want = [float(v1.training_step(vb)) for _ in range(2)]
    v2 = P.VQDIFModel.load_from_checkpoint(vpath)           # the reference's own loading route (hyper_parameters + state_dict)
    v2.make_trainer(dict(lr=1e-3))
    v2.resume(vpath)
    assert v2.trainer.step_count == 2
    assert [float(v2.training_step(vb)) for _ in range(2)] == want
    # StepLR across an in-process resume (vqdif.py:127-133): the base stays optim_opt.lr, the decayed rate is restored once
    v3 = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VQDIF", "kwargs": kw})
    v3.hparams["optim_opt"] = dict(lr=1e-3, scheduler="StepLR", step_size=2, gamma=0.5)
    v3.make_trainer()
    assert v3.on_epoch_end(0) == 1e-3 and v3.on_epoch_end(1) == 5e-4          # epoch 1 ends: (1 + 1) // 2 = 1 decay
    p3 = v3.save_checkpoint(str(tmp_path / "ck" / "vq3.ckpt"), epoch=1)
    v3.resume(p3)
    assert v3.trainer.lr == 5e-4 and v3._lr0 == 1e-3
    assert v3.on_epoch_end(2) == 5e-4 and v3.on_epoch_end(3) == 2.5e-4        # NOT 1.25e-4: the decay is applied to the base once
    # a checkpoint written before the optimizer key was renamed still restores its moments
    ck3 = torch.load(p3, map_location="cpu", weights_only=False)
    ck3["optimizer_states"] = [ck3.pop("sfmi_optimizer_state")]
    torch.save(ck3, p3)
    v3.trainer.step_count = 0
    v3.resume(p3)
    assert v3.trainer.step_count == ck3["global_step"]


def test_rs_ag_trainer_checkpoint_round_trip_through_the_plugin(dev, tmp_path):
    """an rs_ag trainer (sharded AdamW, overlapped parameter gathers) saved and reloaded through ShapeFormerModel.save_checkpoint /
    load_checkpoint.  The rebuilt trainer keeps the old one's settings (gradient mode, fusion, overlap), the rank-local optimizer shard is
    accepted, and the resumed run continues with exactly the losses of the uninterrupted one.  One rank, gloo, collectives forced on
    (the sharded code path with world 1); the callback-level sampling right after a training step drains the in-flight gathers."""
    import torch.distributed as dist
    from shapeformer_amd import plugin as P, synthetic
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/pg", rank=0, world_size=1)
    try:
        b = synthetic.make_batch(9, 2, n_full=8192, n_partial=4096)
        batch = {k: torch.from_numpy(v) for k, v in b.items()}

        def steps(model, n, seed0):
            out = []
            for i in range(n):
                np.random.seed(seed0 + i)
                out.append(float(model.training_step(batch)))
            return out
        opt = P.get_opt(_opt())
        m1 = P.instantiate_from_opt(opt["pl_model_opt"])
        tr = m1.make_trainer(dict(lr=1e-3), dist=dist, grad_sync="rs_ag", single_rank_collectives=True)
        assert tr.buckets.active and tr.buckets.mode == "rs_ag" and tr.settings()["single_rank_collectives"]
        steps(m1, 2, 0)
        assert tr.buckets.params_in_flight(), "the step leaves its parameter gathers to the next forward"
        # a reader that goes straight to the transformer (what VisShapeFormer.compute_batch does) drains them through the model's hook
        enc = m1.pipe.encode_cloud(batch["Xct"][:1].to(dev))
        m1.transformer.sample(enc["c_tokens"], enc["Lc"], max_steps=4, stop_early=False)
        assert not tr.buckets.params_in_flight()
        path = m1.save_checkpoint(str(tmp_path / "ck" / "rsag.ckpt"))
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert ck["sfmi_optimizer_state"]["shard_ranges"] and ck["sfmi_optimizer_state"]["world"] == 1
        want = steps(m1, 2, 2)
        m1.load_checkpoint(path)                         # in place: the old trainer is rebuilt on the new tensors AS IT WAS
        t2 = m1.trainer
        assert t2 is not tr and t2.settings() == tr.settings() and t2.buckets.mode == "rs_ag" and t2.step_count == 2
        assert steps(m1, 2, 2) == want
    finally:
        dist.destroy_process_group()


def test_config1_demo_yaml_through_listdataset_and_callback(dev, tmp_path, monkeypatch):
    """BASELINE config 1 on the GPU box: `configs/demo/demo_vqdif.yaml` (its merged option tree, committed as
    tests/golden/demo_ds/demo_vqdif.yaml by oracle/make_golden_demo.py) through OUR plugin loader -> DataModule -> ListDataset
    (list_dataset.py:13-37) -> VisSparseRecon3D.process (vqdif.py:243-269) must reproduce what the REFERENCE's own ListDataset +
    VisSparseRecon3D.compute_batch returned for the same two demo items (tests/golden/demo_vqdif_ref.npz): code indices, packed
    sparse tokens and occupied-cell mask exactly, logits to fp32 tolerance; and leave the reference's result files behind."""
    from shapeformer_amd import plugin as P
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ds_dir = os.path.join(gold, "demo_ds")
    opt = P.get_opt(os.path.join(ds_dir, "demo_vqdif.yaml"))
    assert opt["pl_model_opt"]["class"] == "shapeformer.models.vqdif.vqdif.VQDIF" and opt["datamodule_opt"]["kwargs"]["test_batch_size"] == 1
    monkeypatch.chdir(ds_dir)                      # the YAML's `ditem_list` is relative to the working directory, as in the reference
    model = P.instantiate_from_opt(opt["pl_model_opt"])
    dm = P.instantiate_from_opt(opt["datamodule_opt"])
    dm.setup("test")
    assert type(dm.test_set).__name__ == "ListDataset" and len(dm.test_set) == 2 and dm.train_set is None
    item = dm.test_set[0]
    assert set(item) == {"Xbd", "Xct"} and item["Xbd"].shape == (2048, 3)          # subsample: False -> the stored points, in order
    cbo = opt["callbacks"]["vis_recon"]
    cbo["kwargs"]["data_dir"] = str(tmp_path / "demo_vqdif")
    cb = P.instantiate_from_opt(cbo)
    assert cb.visual_indices == "all"             # the YAML says `visual_indices: all` (plutil.py:177)
    out = cb.process(model, dm.test_set)
    ref = np.load(os.path.join(gold, "demo_vqdif_ref.npz"))
    Q = int(ref["Q"])
    assert sorted(out) == ["0", "1"]
    worst = 0.0
    for i, name in enumerate(ref["names"]):
        comp = np.load(tmp_path / "demo_vqdif" / "computed" / f"{i}.npy", allow_pickle=True).item()
        assert set(comp) == {"logits", "quant_ind", "sparse", "grid_mask", "batch"}
        assert np.array_equal(comp["quant_ind"].astype(np.int64), ref[f"{name}_quant_ind"].astype(np.int64)), name
        assert np.array_equal(comp["sparse"].astype(np.int64), ref[f"{name}_sparse"].astype(np.int64)), name
        assert np.array_equal(np.packbits(comp["grid_mask"].astype(bool)), ref[f"{name}_grid_mask"]), name
        got, want = comp["logits"].reshape(-1), ref[f"{name}_logits"]
        assert got.shape == (Q ** 3,)
        err = np.abs(got - want)
        worst = max(worst, float(err.max()))
        assert np.all(err <= 2e-4 + 1e-4 * np.abs(want)), (name, float(err.max()))
        assert os.path.exists(tmp_path / "demo_vqdif" / "meshes" / f"{i}.ply") or len(out[str(i)]["recon_mesh"]["face"]) == 0
    print(f"config 1 via YAML -> ListDataset -> VisSparseRecon3D: logits max |diff| vs the reference {worst:.2e}")


def test_b1_vqdif_submodules_from_the_yaml_opts_match_the_reference_vectors(dev, vq16_sd):
    """`shapeformer.models.vqdif.enc.LocalPoolPointnet`, `...quantizer.Quantizer`, `...dec.LocalDecoder`
    resolve under their reference dotted names and are built from the opt dicts of configs/vqdif/shapenet_res16.yaml (via the
    merged tree of the demo YAML, committed under tests/golden/demo_ds), each with the reference's call contract
    (vqdif.py:28-48,60-76): encoder(Xbd / 2) -> (fea, mask); quantizer(fea) -> (q, q_st, idx, diff), get_code(idx);
    decoder(Xtg / 2, c_grid) -> (B,N,1).  Chained exactly as VQDIF.forward chains them they must reproduce what the REFERENCE
    returned on the same weights (tests/golden/vqdif16_small.npz): mask bits, latent taps, code indices, logits."""
    from oracle import vqdif_oracle as O
    from shapeformer_amd import plugin as P, vqdif as V
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    kw = P.get_opt(os.path.join(gold, "demo_ds", "demo_vqdif.yaml"))["pl_model_opt"]["kwargs"]
    z = np.load(os.path.join(gold, "vqdif16_small.npz"))

    def make(o, cls):
        assert P.load_object(o["class"]) is cls
        return P.instantiate_from_opt(dict(o, kwargs=dict(o["kwargs"], state_dict=vq16_sd, device=dev)))
    enc = make(kw["encoder_opt"], V.LocalPoolPointnet)
    qz = make(kw["quantizer_opt"], V.Quantizer)
    dec = make(kw["decoder_opt"], V.LocalDecoder)
    cloud = torch.from_numpy(z["cloud"])
    fea, mask = enc(cloud / 2.0)                                                     # vqdif.py:36
    assert fea.shape == (2, 128, 16, 16, 16) and mask.dtype == torch.bool and mask.shape == (2, 16, 16, 16)
    assert np.array_equal(np.packbits(mask.cpu().numpy()), z["grid_mask"])
    lat = fea.permute(0, 2, 3, 4, 1).cpu()
    scale = float(np.abs(z["latent_sel"]).max())
    np.testing.assert_allclose(lat.reshape(2, -1, 128)[:, ::61].numpy(), z["latent_sel"], atol=2e-5 * scale + 1e-4)
    q, q_st, idx, diff = qz(fea)                                                     # vqdif.py:41-42
    assert idx.dtype == torch.int64 and idx.shape == (2, 16, 16, 16) and q.shape == fea.shape and torch.equal(q, q_st)
    assert np.array_equal(idx.cpu().numpy(), z["quant_ind_raw"].astype(np.int64))
    assert torch.equal(qz.get_code(idx), q) and qz.get_code(idx, bchw=False).shape == (2, 16, 16, 16, 128)
    np.testing.assert_allclose(float(diff), float(((fea - q) ** 2).mean()), rtol=1e-5)
    Q = int(z["Q"])
    Xtg = torch.from_numpy(O.make_grid(Q))[None].expand(2, -1, -1)
    code = qz.get_code(torch.from_numpy(z["quant_ind"].astype(np.int64)))           # decode_index (vqdif.py:74-76)
    lg = dec(Xtg / 2.0, code)                                                        # vqdif.py:71
    assert lg.shape == (2, Q ** 3, 1)
    np.testing.assert_allclose(lg.cpu().numpy()[..., 0], z["logits"], atol=2e-4, rtol=1e-4)
    # a VQDIF composed of these three objects is the model the plugin builds; swapping ONE module keeps the others' weights
    model = P.instantiate_from_opt({"class": "shapeformer.models.vqdif.vqdif.VQDIF", "kwargs": dict(kw, state_dict=vq16_sd, device=dev)})
    assert isinstance(model.encoder, V.LocalPoolPointnet) and isinstance(model.quantizer, V.Quantizer) and isinstance(model.decoder, V.LocalDecoder)
    qi, mode, _ = model.quantize_cloud(cloud)
    assert int(mode) == int(z["mode"]) and np.array_equal(qi.cpu().numpy(), z["quant_ind"].astype(np.int64))
    assert sorted(model.core.state_dict_np()) == sorted(vq16_sd)
    # hyper-parameters outside what the kernels are built for name the limit (never "outside the hot path")
    with pytest.raises(ValueError, match="n_embd=96"):
        V.Quantizer(4096, 96, device=dev)
    with pytest.raises(ValueError, match="hidden_size=256"):
        V.LocalDecoder(**dict(kw["decoder_opt"]["kwargs"], hidden_size=256), device=dev)
    with pytest.raises(ValueError, match="grid_resolution=32"):
        V.LocalPoolPointnet(**dict(kw["encoder_opt"]["kwargs"], grid_resolution=32), device=dev)
