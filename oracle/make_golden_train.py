"""ORACLE tooling (this container only): pins oracle/vqdif_train_oracle.py against the REAL reference training step.

Builds the reference VQDIF (configs/vqdif/shapenet_res16.yaml) with hash weights, puts it in train mode, runs
`get_loss(batch)` + `loss.backward()` on a small seeded batch and compares losses, every parameter gradient and the
EMA-updated codebook buffers with the oracle.  Writes tests/golden/vqdif_train.npz (inputs + per-tensor gradient
checksums + losses + EMA checksums) for the GPU parity test.

    python oracle/make_golden_train.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport, vqdif_oracle as VO, vqdif_train_oracle as TO  # noqa: E402


def batch(seed=7, B=1, T=2048, Nt=512):
    rs = np.random.RandomState(seed)
    u = rs.randn(B, T, 3)
    u /= np.linalg.norm(u, axis=-1, keepdims=True)
    Xbd = (u * np.array([0.6, 0.35, 0.45]) + 0.02 * rs.randn(B, T, 3)).clip(-0.95, 0.95).astype(np.float32)
    Xtg = rs.uniform(-1, 1, (B, Nt, 3)).astype(np.float32)
    inside = ((Xtg / np.array([0.6, 0.35, 0.45])) ** 2).sum(-1, keepdims=True) < 1
    return Xbd, Xtg, inside.astype(np.float32)


def main():
    refimport.setup()
    torch.manual_seed(0)
    model = refimport.build_vqdif(16)
    model.train()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    Xbd, Xtg, Ytg = batch(Nt=4096)
    # keep 512 query points whose MLP pre-activations are all >= 2e-3 away from a ReLU kink (see sdf_head_margin)
    with torch.no_grad():
        tsd = {k: v for k, v in sd.items()}
        idx = TO.training_losses(tsd, torch.from_numpy(Xbd), torch.from_numpy(Xtg[:, :8]), torch.from_numpy(Ytg[:, :8]), 0.001)["idx"]
        grid = VO.decoder_grid(tsd, VO.get_code(tsd, idx.view(1, 16, 16, 16)))
        margin = TO.sdf_head_margin(tsd, grid, torch.from_numpy(Xtg))[0].numpy()
    keep = np.nonzero(margin > 2e-3)[0][:512]
    assert len(keep) == 512, len(keep)
    print("query points kept:", len(keep), "of 4096; min margin", float(margin[keep].min()))
    Xtg, Ytg = Xtg[:, keep], Ytg[:, keep]
    b = {k: torch.from_numpy(v) for k, v in dict(Xbd=Xbd, Xtg=Xtg, Ytg=Ytg).items()}
    beta = float(model.criterion.beta)
    t0 = time.time()
    losses = model.get_loss(b, 0)
    losses["loss"].backward()
    print(f"reference fwd+bwd {time.time() - t0:.1f}s  loss {float(losses['loss']):.6f} recon {float(losses['recon_loss']):.6f} diff {float(losses['diff_loss']):.6f} beta {beta}")
    ref_g = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    t0 = time.time()
    out, g = TO.loss_and_grads(sd, b["Xbd"], b["Xtg"], b["Ytg"], beta)
    print(f"oracle fwd+bwd {time.time() - t0:.1f}s  loss {float(out['loss']):.6f}")
    assert abs(float(out["loss"]) - float(losses["loss"])) < 1e-6
    worst = 0.0
    assert set(ref_g) == {k for k, v in g.items() if v is not None}, set(ref_g) ^ {k for k, v in g.items() if v is not None}
    for k, rg in ref_g.items():
        e = float((g[k] - rg).abs().max() / (rg.abs().max() + 1e-12))
        worst = max(worst, e)
    print("worst relative gradient error oracle vs reference:", worst)
    assert worst < 1e-4
    N, z, emb = TO.ema_update(sd, out["x"], out["idx"])
    msd = model.state_dict()
    for nm, t in (("N", N), ("z_avg", z), ("embedding.weight", emb)):
        ref = msd["quantizer." + nm]
        e = float(((t - ref).abs() / (ref.abs() + 1e-3)).max())
        print("EMA", nm, "max rel diff", e, " max |ref|", float(ref.abs().max()))
        assert e < 1e-5
    fx = dict(Xbd=Xbd, Xtg=Xtg, Ytg=Ytg, beta=np.float32(beta), loss=np.float32(losses["loss"].item()),
              recon_loss=np.float32(losses["recon_loss"].item()), diff_loss=np.float32(losses["diff_loss"].item()),
              idx=out["idx"].numpy().astype(np.int16))
    names = sorted(ref_g)
    fx["grad_names"] = np.array(names)
    fx["grad_sum"] = np.array([float(ref_g[k].double().sum()) for k in names])
    fx["grad_abs"] = np.array([float(ref_g[k].double().abs().sum()) for k in names])
    fx["grad_max"] = np.array([float(ref_g[k].abs().max()) for k in names])
    fx["ema_N_sum"], fx["ema_emb_abs"] = np.float64(msd["quantizer.N"].double().sum()), np.float64(msd["quantizer.embedding.weight"].double().abs().sum())
    path = os.path.join(ROOT, "tests", "golden", "vqdif_train.npz")
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
