"""GPU: the reference's YAML/plugin surface end to end — instantiate_from_opt on a config with the shipped YAML's keys
(shapenet_scale.yaml layout, transformer shrunk for test speed), completion and one training step."""
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
            mask_invalid_completion=True,
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
