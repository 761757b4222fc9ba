"""ORACLE (test infrastructure only): the VQDIF training step restated on the functional oracle.

  training_losses   VQDIF.forward in training mode + VQLoss (vqdif.py:78-98,151-167; quantizer.py:31-89): encode ->
                    nearest code (pre-update codebook) -> straight-through -> decode at Xtg -> BCE-with-logits (mean)
                    + beta * mean((x - q.detach())^2)
  ema_update        quantizer.py:68-86 (gamma .99, eps 1e-7): N, z_avg, embedding.weight after one training forward
  loss_and_grads    torch autograd of training_losses w.r.t. every trainable tensor
Pinned against the imported reference (real `VQDIF.get_loss(...).backward()` in train mode on hash weights) by
oracle/make_golden_train.py; fixture tests/golden/vqdif_train.npz.
"""
import torch
import torch.nn.functional as F

from . import vqdif_oracle as VO


def training_losses(sd, Xbd, Xtg, Ytg, beta):
    fea, _ = VO.encode(sd, Xbd)                                  # (B,d,R,R,R)
    B, d = fea.shape[:2]
    W = sd["quantizer.embedding.weight"]
    x = fea.permute(0, 2, 3, 4, 1).contiguous().view(-1, d)
    with torch.no_grad():
        dist = (x ** 2).sum(1, keepdim=True) - 2 * torch.mm(x, W.t()) + (W.t() ** 2).sum(0, keepdim=True)
        idx = torch.max(-dist, dim=1)[1]
    q = W.detach()[idx].view(B, *fea.shape[2:], d).permute(0, 4, 1, 2, 3).contiguous()
    q_st = (q - fea).detach() + fea
    diff = (fea - q.detach()).pow(2).mean()
    logits = VO.sdf_query(sd, VO.decoder_grid(sd, q_st), Xtg)
    recon = F.binary_cross_entropy_with_logits(logits, Ytg)
    return dict(loss=recon + beta * diff, recon_loss=recon, diff_loss=diff, idx=idx, x=x.detach(), logits=logits.detach())


def ema_update(sd, x, idx, gamma=0.99, eps=1e-7):
    """-> (N, z_avg, embedding.weight) after the update; x (n,d) detached latent rows, idx (n,) chosen codes."""
    K = sd["quantizer.embedding.weight"].shape[0]
    onehot = F.one_hot(idx, K).to(x.dtype)
    N = sd["quantizer.N"] * gamma + (1 - gamma) * onehot.sum(0)
    z = sd["quantizer.z_avg"] * gamma + (1 - gamma) * torch.mm(x.t(), onehot).t()
    n = N.sum()
    w = (N + eps) / (n + K * eps) * n
    return N, z, z / w.unsqueeze(1)


def trainable_keys(sd):
    return [k for k in sd if not k.startswith("quantizer.")]


def loss_and_grads(sd, Xbd, Xtg, Ytg, beta, relu_masks=None, pool_index=None):
    """relu_masks: optional list of boolean tensors, one per ReLU of the forward in call order (conv outputs channels-last):
    the forward then uses x * mask instead of relu(x) (vqdif_oracle.RELU_MASKS) - the activation pattern of ANOTHER
    implementation's forward, so that both differentiate the same piecewise-linear function.  pool_index: likewise the selected
    window element (0..7, channels-last) of every 2^3 max-pool (vqdif_oracle.POOL_INDEX)."""
    keys = trainable_keys(sd)
    leaf = dict(sd)
    for k in keys:
        leaf[k] = sd[k].clone().requires_grad_(True)
    if relu_masks is not None:
        VO.RELU_MASKS = [torch.as_tensor(m) for m in relu_masks]
    if pool_index is not None:
        VO.POOL_INDEX = [torch.as_tensor(m) for m in pool_index]
    try:
        out = training_losses(leaf, Xbd, Xtg, Ytg, beta)
        assert not VO.RELU_MASKS, f"{len(VO.RELU_MASKS)} ReLU masks were not consumed"
        assert not VO.POOL_INDEX, f"{len(VO.POOL_INDEX)} pool index tensors were not consumed"
    finally:
        VO.RELU_MASKS = VO.POOL_INDEX = None
    grads = torch.autograd.grad(out["loss"], [leaf[k] for k in keys], allow_unused=True)
    return out, {k: g for k, g in zip(keys, grads)}


def sdf_head_margin(sd, grid, Xtg):
    """Per query point: the smallest |pre-activation| over every ReLU of the implicit decoder's MLP (dec.py:88-100).
    Two fp32 implementations agree to ~1e-5 on these values; a point whose margin is below that can take the other
    branch of a ReLU, which changes ITS whole gradient contribution (1/N of every upstream gradient).  The parity
    fixture therefore keeps only points with a margin far above the fp32 noise."""
    with torch.no_grad():
        p = (Xtg / 2.0).float()
        c = VO.trilinear_sample(grid, p)
        net = F.linear(p, sd["decoder.fc_p.weight"], sd["decoder.fc_p.bias"])
        m = torch.full(net.shape[:2], float("inf"))
        for i in range(5):
            net = net + F.linear(c, sd[f"decoder.fc_c.{i}.weight"], sd[f"decoder.fc_c.{i}.bias"])
            m = torch.minimum(m, net.abs().min(-1)[0])
            h = F.linear(F.relu(net), sd[f"decoder.blocks.{i}.fc_0.weight"], sd[f"decoder.blocks.{i}.fc_0.bias"])
            m = torch.minimum(m, h.abs().min(-1)[0])
            net = net + F.linear(F.relu(h), sd[f"decoder.blocks.{i}.fc_1.weight"], sd[f"decoder.blocks.{i}.fc_1.bias"])
        return torch.minimum(m, net.abs().min(-1)[0])
