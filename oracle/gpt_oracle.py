"""CPU oracle for the CondTupleGPT / ShapeFormer sampling half of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/vqdif_oracle.py header).  Functional torch-CPU fp32
restatement of shapeformer/models/shapeformer/transformer/mingpt.py:46-111,256-319 and
shapeformer/models/shapeformer/shapeformer.py:54-123 over a {key: tensor} state dict with the
reference's key names.  Pinned against the imported reference by oracle/make_golden.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import tokens_oracle as T


class GPTCfg:
    def __init__(self, n_embd=1024, n_head=16, n_layers=(20, 4), block_size=812,
                 vocab_sizes=(4097, 4097), end_tokens=(4096, 4096)):
        self.n_embd, self.n_head, self.n_layers = n_embd, n_head, tuple(n_layers)
        self.block_size, self.vocab_sizes, self.end_tokens = block_size, tuple(vocab_sizes), tuple(end_tokens)


def embeddings(sd, idx, extra, L_cond):
    """mingpt.py:256-286: tok_embs[0][pos] + tok_embs[1][val] + extra_tok_embs[0][extra] + pos."""
    L = idx.shape[1]
    x = F.embedding(idx[..., 0], sd["tok_embs.0.weight"]) + F.embedding(idx[..., 1], sd["tok_embs.1.weight"])
    x = x + F.embedding(extra[..., 0], sd["extra_tok_embs.0.weight"])
    pos = torch.cat([sd["cond_pos_emb"][:, :L_cond], sd["pos_emb"][:, :L - L_cond]], dim=1)
    return x + pos


def dropout_mask(key, site, shape, p):
    """nn.Dropout(p) as an explicit multiplier tensor: element i (C order) is 0 iff hash_unit("dropout-<key>-<site>")[i] < p,
    else 1/(1-p) - the counter-hash masks the HIP training kernels apply (csrc/sfmi_common.h:sfmi_dropout_mul).  torch's
    own Philox stream cannot be reproduced on the device, so both sides use this one (same Bernoulli(1-p) distribution)."""
    from shapeformer_amd.weights import hash_unit
    n = int(np.prod(shape))
    u = hash_unit(f"dropout-{key}-{site}", n).reshape(shape)
    return torch.from_numpy(np.where(u < np.float32(p), np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(p))).astype(np.float32))


def block(sd, p, x, n_head, kv=None, drop=None):
    """mingpt.py:46-111.  drop=None: eval (dropout = identity); drop=(key, site_prefix, p_resid, p_attn): train mode with the
    explicit masks of `dropout_mask`.  With kv=(K,V) caches: x holds only the new positions; returns the updated caches (used
    by the KV-cached baseline, equivalent by A11)."""
    B, Tn, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "ln1.weight"], sd[p + "ln1.bias"], 1e-5)
    k = F.linear(h, sd[p + "attn.key.weight"], sd[p + "attn.key.bias"]).view(B, Tn, n_head, C // n_head).transpose(1, 2)
    q = F.linear(h, sd[p + "attn.query.weight"], sd[p + "attn.query.bias"]).view(B, Tn, n_head, C // n_head).transpose(1, 2)
    v = F.linear(h, sd[p + "attn.value.weight"], sd[p + "attn.value.bias"]).view(B, Tn, n_head, C // n_head).transpose(1, 2)
    if kv is not None and kv[0] is not None:
        k = torch.cat([kv[0], k], dim=2)
        v = torch.cat([kv[1], v], dim=2)
    Tk = k.shape[2]
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(k.size(-1)))
    causal = torch.tril(torch.ones(Tk, Tk, dtype=torch.bool))[Tk - Tn:, :]
    att = att.masked_fill(~causal, float("-inf"))
    att = F.softmax(att, dim=-1)
    if drop is not None and drop[3] > 0:
        att = att * dropout_mask(drop[0], drop[1] + ".attn", att.shape, drop[3])            # mingpt.py:85
    y = (att @ v).transpose(1, 2).contiguous().view(B, Tn, C)
    a = F.linear(y, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    if drop is not None and drop[2] > 0:
        a = a * dropout_mask(drop[0], drop[1] + ".proj", a.shape, drop[2])                  # mingpt.py:90
    x = x + a
    h = F.layer_norm(x, (C,), sd[p + "ln2.weight"], sd[p + "ln2.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"]))
    m = F.linear(h, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    if drop is not None and drop[2] > 0:
        m = m * dropout_mask(drop[0], drop[1] + ".mlp", m.shape, drop[2])                   # mingpt.py:105
    x = x + m
    return x, (k, v)


def head(sd, s, x):
    """heads[s] = LayerNorm -> Linear(no bias) (mingpt.py:221-230)."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[f"heads.{s}.0.weight"], sd[f"heads.{s}.0.bias"], 1e-5)
    return F.linear(h, sd[f"heads.{s}.1.weight"])


def stage(sd, cfg, s, x, caches=None, dropout=None):
    new = []
    for n in range(cfg.n_layers[s]):
        li = n + (cfg.n_layers[0] if s else 0)          # site names use the flat layer index, as the HIP trainer does
        drop = None if dropout is None else (dropout["key"], f"L{li}", dropout["p"][1], dropout["p"][2])
        x, kv = block(sd, f"blocks.{s}.{n}.", x, cfg.n_head, None if caches is None else caches[n], drop)
        new.append(kv)
    return x, new


def forward_logits(sd, cfg, idx, extra, L_cond, target_idx, dropout=None):
    """CondTupleGPT.forward / compute_logits (mingpt.py:287-296,311-319): teacher-forced logits.
    dropout=dict(key=str, p=(embd_pdrop, resid_pdrop, attn_pdrop)): train mode (drops[i] on each stage input, :292)."""
    x = embeddings(sd, idx, extra, L_cond)
    if dropout is not None and dropout["p"][0] > 0:
        x = x * dropout_mask(dropout["key"], "emb0", x.shape, dropout["p"][0])
    x, _ = stage(sd, cfg, 0, x, dropout=dropout)
    l0 = head(sd, 0, x)
    x = x + F.embedding(target_idx[..., 0], sd["tok_embs.0.weight"])
    if dropout is not None and dropout["p"][0] > 0:
        x = x * dropout_mask(dropout["key"], "emb1", x.shape, dropout["p"][0])
    x, _ = stage(sd, cfg, 1, x, dropout=dropout)
    l1 = head(sd, 1, x)
    return [l0, l1]


def training_loss(sd, cfg, c_indices, z_indices, extra, dropout=None):
    """ShapeFormer.forward + shared_step (shapeformer.py:26-46,132-140): mean of the two CEs over
    outputs from index L_c-1 on, end-token padding included."""
    cz = torch.cat([c_indices, z_indices], dim=1)
    L_c = c_indices.shape[1]
    logits = forward_logits(sd, cfg, cz[:, :-1], extra[:, :-1], L_c, cz[:, 1:], dropout=dropout)
    loss = 0
    for i in range(2):
        lg = logits[i][:, L_c - 1:, :]
        loss = loss + F.cross_entropy(lg.reshape(-1, lg.shape[-1]), z_indices[..., i].reshape(-1))
    return loss / 2


def uniforms(seed, n_steps, B):
    """Counter-hash uniforms u[step, tuple_i, row] in [0,1) shared by the oracle and the HIP
    sampler (csrc/sampling.hip sf_uniform) — replaces torch.multinomial's RNG stream."""
    from shapeformer_amd.weights import hash_unit
    return hash_unit(f"sample-uniforms-{seed}", n_steps * 2 * B).reshape(n_steps, 2, B)


@torch.no_grad()
def sample_indices(sd, cfg, c_indices, max_steps, u, top_k=100, top_p=0.4, temperature=1.0,
                   best_in_first=True, mask_invalid=True, mask_invalid_completion=True,
                   use_cache=True, stop_early=True, force_tokens=None, return_logits=True, z_indices=None):
    """ShapeFormer.sample_indices (shapeformer.py:54-123) with AR_N extra indices
    (representers.py:188-196), the representer's sampling_masker (:120-155) and
    filter_sampling_logits (common.py:260-285); multinomial -> inverse CDF on supplied u.

    use_cache=False recomputes the whole prefix every step exactly like the reference;
    use_cache=True is the KV-cached equivalent.
    force_tokens (B,steps,2): teacher-forced stepwise mode — logits are recorded, tokens forced.
    Steps are capped so the sequence never exceeds block_size (the reference's crop at
    shapeformer.py:73-76 is buggy/unreachable; the build stops instead).
    z_indices (B,L_z,2): tokens already generated (shapeformer.py:60-70 copies cat(c, z) into `sampled` and continues after
    them); the step counter j of the masker / history / uniforms starts at 0 at the first NEW token, as in the reference's loop,
    and the returned tokens are sampled[:, L_c:] - the prefix included (shapeformer.py:121).
    """
    c = torch.as_tensor(c_indices).long()
    B, L_c, _ = c.shape
    z = torch.zeros(B, 0, 2, dtype=torch.long) if z_indices is None else torch.as_tensor(z_indices).long()
    L_z = z.shape[1]
    end = cfg.end_tokens
    max_steps = min(max_steps, cfg.block_size - L_c - L_z)
    sampled = torch.zeros(B, L_c + L_z + max_steps, 2, dtype=torch.long)
    sampled[:, :L_c] = c
    sampled[:, L_c:L_c + L_z] = z
    hist = [[], []]
    tail = L_c + L_z
    caches = [None, None]
    done_steps = 0
    for j in range(max_steps):
        cz = sampled[:, :tail]
        extra = torch.from_numpy(T.extra_indices_AR_N(cz[:, :L_c].numpy(), cz[:, L_c:].numpy(), end[0]))
        if use_cache:
            lo = 0 if j == 0 else tail - 1
            x = embeddings(sd, cz, extra, L_c)[:, lo:]
            x, caches[0] = stage(sd, cfg, 0, x, caches[0] if j else [None] * cfg.n_layers[0])
        else:
            x = embeddings(sd, cz, extra, L_c)
            x, _ = stage(sd, cfg, 0, x)
        logits = head(sd, 0, x[:, -1:])[:, 0]
        for i in range(2):
            idx_view = sampled[:, :tail + 1].numpy()
            ml = T.sampling_masker(logits.numpy(), idx_view, L_c, j, i, end, mask_invalid,
                                   mask_invalid_completion)
            if return_logits:
                hist[i].append(ml.copy())
            new = np.empty(B, np.int64)
            for b in range(B):
                if force_tokens is not None:
                    new[b] = int(force_tokens[b, j, i])
                elif best_in_first and b == 0:
                    new[b] = int(np.argmax(ml[b]))  # top_k=1/top_p=.001 == argmax (shapeformer.py:98-101)
                else:
                    f = T.filter_sampling_logits(ml[b], top_k, top_p, temperature)
                    new[b] = T.sample_filtered(f, u[j, i, b])
            sampled[:, tail, i] = torch.from_numpy(new)
            if i == 1:
                break
            tgt = sampled[:, 1:tail + 1, 0]
            if use_cache:
                x1 = x + F.embedding(tgt[:, -x.shape[1]:], sd["tok_embs.0.weight"])
                x1, caches[1] = stage(sd, cfg, 1, x1, caches[1] if j else [None] * cfg.n_layers[1])
            else:
                x1 = x + F.embedding(tgt, sd["tok_embs.0.weight"])
                x1, _ = stage(sd, cfg, 1, x1)
            logits = head(sd, 1, x1[:, -1:])[:, 0]
        tail += 1
        done_steps = j + 1
        # shapeformer.py:112-115: a row has stopped once ANY tuple element equals its end token
        if stop_early and bool((sampled[:, tail - 1] == torch.tensor(end)).any(-1).all()):
            break
    out = sampled[:, L_c:tail].numpy()
    if return_logits:
        hist = [np.stack(h, axis=1) for h in hist]
    return out, hist, done_steps


def compute_log_probs(samples, logits_history):
    """shapeformer.py:407-418: log-softmax of the (masked) logits at the sampled token."""
    S, L, tn = samples.shape
    out = np.zeros(samples.shape)
    for ti in range(tn):
        lg = np.asarray(logits_history[ti], dtype=np.float64)
        m = lg.max(-1, keepdims=True)
        lse = m + np.log(np.exp(lg - m).sum(-1, keepdims=True))
        lp = lg - lse
        out[:, :, ti] = np.take_along_axis(lp, samples[:, :, ti][..., None], axis=-1)[..., 0]
    return out
