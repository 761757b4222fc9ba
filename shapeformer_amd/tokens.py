"""Sparse (pos,code) token packing over libsfmi (mirrors shapeformer/models/common.py:84-189).

Device representation is ragged-in-a-padded-buffer: (B,Lpad,2) int32 + len (B,) int32 (see
csrc/tokens.hip).  The `*_ref` helpers return the reference's exact tensor shapes/dtypes
((B,L,2) int64 etc.) and therefore sync once to size the result, as the reference does.
"""
from __future__ import annotations

import torch

from . import _lib as L


def mode_i32(idx, vocab, rows=1):
    """rows=1: whole-tensor mode (reference batch semantics); rows=B: one mode per shape."""
    dev = idx.device
    hist = torch.empty(rows * vocab, device=dev, dtype=torch.int32)
    out = torch.empty(rows, device=dev, dtype=torch.int32)
    L.check(L.lib().sfmi_mode_i32(L.ptr(idx), idx.numel(), vocab, rows, L.ptr(hist), L.ptr(out), L.stream_ptr()), "sfmi_mode_i32")
    return out


def dense2sparse_dev(q, mode, max_length, end_tokens, Lpad=None, tokens=None, length=None):
    """q (B,R,R,R) int32 on device, mode (1,) int32 -> tokens (B,Lpad,2) int32, len (B,) int32."""
    if not q.is_cuda:
        raise L.SfmiError("device tensor required")
    q = q.contiguous()
    B = q.shape[0]
    ncell = q.numel() // B
    Lpad = Lpad or max_length
    tokens = tokens if tokens is not None else torch.empty(B, Lpad, 2, device=q.device, dtype=torch.int32)
    length = length if length is not None else torch.empty(B, device=q.device, dtype=torch.int32)
    L.check(L.lib().sfmi_dense2sparse_i32(L.ptr(q), L.ptr(mode), int(mode.numel() > 1), L.ptr(tokens), L.ptr(length), B, ncell, Lpad, max_length,
                                          int(end_tokens[0]), int(end_tokens[1]), L.stream_ptr()), "sfmi_dense2sparse_i32")
    return tokens, length


def sparse2dense_dev(tokens, length, empty, dense_res, end_tokens, dim=3, out=None, start=None):
    """tokens (B,Lpad,2) int32, len (B,) or None, empty (1,) or (B,) int32, optional per-row start -> dense (B,R,R,R) int32."""
    tokens = tokens.contiguous()
    B, Lpad, _ = tokens.shape
    ncell = dense_res ** dim
    out = out if out is not None else torch.empty((B,) + (dense_res,) * dim, device=tokens.device, dtype=torch.int32)
    L.check(L.lib().sfmi_sparse2dense_i32(L.ptr(tokens), L.ptr(start), L.ptr(length), L.ptr(empty), int(empty.numel() > 1), L.ptr(out), B, ncell, Lpad,
                                          int(end_tokens[0]), int(end_tokens[1]), L.stream_ptr()), "sfmi_sparse2dense_i32")
    return out


def batch_dense2sparse(indices, max_length=None, end_tokens=(100, 200), vocab=4097):
    """common.py:151-168 (unpack=True): (B,R,R,R) -> ((B,L,2) int64 padded with end tokens, mode)."""
    q = indices.to(torch.int32).contiguous()
    B = q.shape[0]
    ncell = q.numel() // B
    mode = mode_i32(q, vocab)
    ml = max_length if max_length is not None else ncell + 1
    tok, ln = dense2sparse_dev(q, mode, ml, end_tokens, Lpad=min(ml, ncell + 1))
    Lmax = int(ln.max().item())  # the reference sizes its output the same way (host sync)
    return tok[:, :Lmax].long(), mode.long()[0]


def batch_sparse2dense_padded(sparse, empty_ind, dense_res, end_tokens):
    """pack_sparse + batch_sparse2dense (common.py:126-140,171-189) on a padded (B,L,2) tensor."""
    tok = sparse.to(torch.int32).contiguous()
    empty = torch.as_tensor([int(empty_ind)], device=tok.device, dtype=torch.int32)
    return sparse2dense_dev(tok, None, empty, dense_res, end_tokens).long()
