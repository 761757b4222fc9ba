"""CPU oracle (numpy, integer-exact) for the sparse (pos,code) token path.

TEST INFRASTRUCTURE ONLY (see oracle/vqdif_oracle.py header).  Restates
shapeformer/models/common.py:84-189, representers.py:120-155,188-196,432-442 and
the deterministic part of common.py:260-299.  Pinned by oracle/make_golden.py
against the imported reference and the known-answer cases of SURVEY.md §4.
"""
from __future__ import annotations

import numpy as np


def get_mode(a):
    """models/common.py:20-23 / torch.mode: most frequent value, smallest on ties."""
    vals, counts = np.unique(np.asarray(a).reshape(-1), return_counts=True)
    return int(vals[np.argmax(counts)])


def dense2packed(indices):
    """common.py:151-163 (unpack=False): rows [b, pos, val] of all cells != mode, row-major."""
    indices = np.asarray(indices)
    B = indices.shape[0]
    mode = get_mode(indices)
    flat = indices.reshape(B, -1)
    b, pos = np.nonzero(flat != mode)
    return np.stack([b, pos, flat[b, pos]], axis=-1).astype(np.int64), mode


def unpack_sparse(packed, max_length=None, end_tokens=(100, 200), batch_size=None):
    """common.py:84-123: (K,3) -> (B, Lmax+1, 2) padded with end tokens; truncate to max_length
    forcing the last column to end tokens.

    Deviation (documented in DESIGN.md): rows are indexed by batch id, and `batch_size` may be
    given so items with zero tokens keep an (all-end-token) row; the reference sizes the output by
    the number of distinct consecutive ids and breaks on empty items.
    """
    packed = np.asarray(packed, dtype=np.int64).reshape(-1, 3)
    b = packed[:, 0]
    B = int(batch_size) if batch_size is not None else (int(len(np.unique(b))) if len(b) else 0)
    counts = np.bincount(b, minlength=B) if len(b) else np.zeros(B, np.int64)
    L = int(counts.max()) + 1 if B else 1
    out = np.empty((B, L, 2), np.int64)
    out[:] = np.asarray(end_tokens, np.int64)[None, None, :]
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]) if B else np.zeros(0, np.int64)
    r = np.arange(len(b)) - starts[b] if len(b) else np.zeros(0, np.int64)
    out[b, r, 0] = packed[:, 1]
    out[b, r, 1] = packed[:, 2]
    if max_length is not None and out.shape[1] > max_length:
        out = out[:, :max_length, :].copy()
        out[:, max_length - 1, :] = np.asarray(end_tokens, np.int64)
    return out


def batch_dense2sparse(indices, max_length=None, end_tokens=(100, 200)):
    """common.py:151-168 (unpack=True) -> ((B,L,2), mode)."""
    packed, mode = dense2packed(indices)
    return unpack_sparse(packed, max_length, end_tokens, batch_size=np.asarray(indices).shape[0]), mode


def pack_sparse(sparse, end_tokens=(100, 200)):
    """common.py:126-140: keep rows where BOTH elements differ from their end token."""
    sparse = np.asarray(sparse, dtype=np.int64)
    keep = (sparse != np.asarray(end_tokens, np.int64)[None, None, :]).all(-1)
    b, l = np.nonzero(keep)
    return np.stack([b, sparse[b, l, 0], sparse[b, l, 1]], axis=-1)


def batch_sparse2dense(packed, empty_ind, dense_res, batch_size=None, dim=3):
    """common.py:171-189: dense filled with empty_ind, dense[b, unravel(pos)] = val."""
    packed = np.asarray(packed, dtype=np.int64).reshape(-1, 3)
    B = int(batch_size) if batch_size is not None else int(len(np.unique(packed[:, 0])))
    dense = np.full((B, dense_res ** dim), int(empty_ind), np.int64)
    dense[packed[:, 0], packed[:, 1]] = packed[:, 2]
    return dense.reshape((B,) + (dense_res,) * dim)


def get_next_cond(c_pos, z_pos, end_token):
    """representers.py:432-442: first cond pos > z pos (searchsorted right); end -> end."""
    c_pos, z_pos = np.asarray(c_pos), np.asarray(z_pos)
    out = np.empty_like(z_pos)
    for b in range(z_pos.shape[0]):
        ids = np.searchsorted(c_pos[b], z_pos[b], side="right")
        ids[z_pos[b] == end_token] = c_pos.shape[1] - 1
        out[b] = c_pos[b][ids]
    out[z_pos == end_token] = end_token
    return out


def extra_indices_AR_N(c_indices, z_indices, end_token):
    """representers.py:188-196: cond tokens -> own pos; generated -> next cond pos."""
    c_extra = c_indices[..., 0]
    z_extra = get_next_cond(c_indices[..., 0], z_indices[..., 0], end_token)
    return np.concatenate([c_extra, z_extra], axis=1)[..., None]


def sampling_masker(logits, idx, L_cond, step_j, tuple_i, end_tokens, mask_invalid=True,
                    mask_invalid_completion=False):
    """representers.py:120-155.  logits (B,V) float; idx (B, seq_tail+1, 2) int (last = being sampled)."""
    logits = np.array(logits, dtype=np.float32, copy=True)
    idx = np.asarray(idx)
    B, V = logits.shape
    latest = idx[:, -2, 0]
    if tuple_i == 1:
        end_mask = idx[:, -1, 0] == end_tokens[0]
        logits[end_mask, :] = -np.inf
        logits[end_mask, end_tokens[1]] = 1.0
        return logits
    positions = np.arange(V)[None, :]
    if mask_invalid and step_j > 0:
        inv = positions <= latest[:, None]
        inv[:, end_tokens[0]] = False
        logits[inv] = -np.inf
    if mask_invalid_completion:
        cond = idx[:, :L_cond, 0]
        cond = np.concatenate([cond, np.full((B, 1), 1 + end_tokens[0], cond.dtype)], axis=1)
        for b in range(B):
            nid = np.searchsorted(cond[b], latest[b], side="right")
            logits[b, positions[0] > cond[b][nid]] = -np.inf
    return logits


def filter_sampling_logits(logits, top_k, top_p, temperature):
    """common.py:260-285 on one row: /T; keep >= k-th largest; sort desc; drop where
    cumsum(softmax) > p shifted right by one."""
    l = np.array(logits, dtype=np.float32, copy=True) / np.float32(temperature)
    V = l.shape[-1]
    top_k = min(top_k, V)
    if top_k > 0:
        kth = np.sort(l)[::-1][top_k - 1]
        l[l < kth] = -np.inf
    if top_p > 0.0:
        order = np.argsort(-l, kind="stable")
        sl = l[order]
        e = np.exp(sl - sl[0], dtype=np.float32)
        cum = np.cumsum(e / e.sum(dtype=np.float32), dtype=np.float32)
        rem = cum > np.float32(top_p)
        rem[1:] = rem[:-1].copy()
        rem[0] = False
        l[order[rem]] = -np.inf
    return l


def sample_filtered(filtered, u):
    """Inverse-CDF draw from softmax(filtered) over candidates ordered by (logit desc, index asc).

    Stands in for torch.multinomial(probs, 1) (common.py:295-297): same distribution, but a
    deterministic function of the supplied uniform u in [0,1) so GPU and oracle can be compared.
    Convention (shared with csrc/sampling.hip): e_i = exp(l_i - l_max) in f32, running f32 sum in
    candidate order, pick first i with csum_i > u * total; fall back to the last candidate.
    """
    l = np.asarray(filtered, dtype=np.float32)
    order = np.argsort(-l, kind="stable")
    sl = l[order]
    m = int(np.isfinite(sl).sum())
    e = np.exp(sl[:m] - sl[0], dtype=np.float32)
    cs = np.cumsum(e, dtype=np.float32)
    t = np.float32(u) * cs[-1]
    i = int(np.searchsorted(cs, t, side="right"))
    return int(order[min(i, m - 1)])
