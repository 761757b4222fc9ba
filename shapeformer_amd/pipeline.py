"""Shape-completion driver: partial cloud -> VQDIF tokens -> AR sampling -> dense code grid -> occupancy.

The compute half of the reference's inference drivers, as plain functions returning the same dict keys:
`VisShapeFormer.compute_batch` + `vis_ind`/`decode_sample_indices` (shapeformer.py:222-260,332-391) and
`VisSparseRecon3D.compute_batch` (vqdif.py:243-269).  Rendering / mesh IO are out of scope (SURVEY §8 f).
Everything stays on the device between stages; every stage is a libsfmi (HIP) call.
"""
from __future__ import annotations

import torch

from . import tokens as T


def default_chains(B):
    """Number of interleaved hipGraph decode chains (gpt.py:sample_microbatched) for B rows: <= 64 rows per chain below 192 rows
    (2 chains already for 32..64 rows); from 192 rows on FOUR chains - one per hardware queue of the runtime (more queues than 4,
    GPU_MAX_HW_QUEUES=8, or 6 chains are slower; profiles/r02_decode_step_experiments.md).  4 x 48 rows: 4.19 ms/step against 4.29
    for 3 x 64; 4 x 80 rows (320 shapes) is the best rows-per-launch trade measured: 78.8 shapes/s against 76.0 at 192."""
    if B >= 192:
        return 4                        # 4 x 96 rows measured best; a chain holds up to 192 rows, more than 768 rows run as successive rounds (gpt.sample_microbatched)
    return -(-B // 64) if B > 64 else (2 if B >= 32 else 1)


class ShapeCompletion:
    def __init__(self, vq, gpt, voxel_res=16, block_size=812, end_tokens=(4096, 4096)):
        self.vq, self.gpt = vq, gpt
        self.R, self.end = voxel_res, tuple(end_tokens)
        self.max_length = block_size // 2  # representers.py:65

    @torch.no_grad()
    def encode_cloud(self, cloud):
        """ShapeRepresenter.encode_cloud (representers.py:69-77), one empty code per shape (batch-1 semantics)."""
        q, mode, raw, mask, latent = self.vq.quantize_cloud_dev(cloud, per_shape_mode=True)
        B = q.shape[0]
        mode2 = T.mode_i32(q, self.vq.K + 1, rows=B)  # batch_dense2sparse recomputes the mode (common.py:155)
        tok, ln = T.dense2sparse_dev(q, mode2, self.max_length, self.end, Lpad=self.max_length)
        return dict(quant_ind=q, empty_index=mode2, c_tokens=tok, Lc=ln, grid_mask=mask)

    @torch.no_grad()
    def complete(self, Xct, max_steps=512, decode_res=128, top_k=100, top_p=0.4, temperature=1.0, seed=0,
                 best_in_first=False, stop_early=True, sigmoid=True, mask_invalid=True, mask_invalid_completion=True,
                 n_micro=None, timings=None):
        """One (pos,val) sequence per input cloud -> dict(samples tokens, dense code grid, occupancy (B,Q^3)).
        timings: optional dict filled with stage times in ms (encode / prefill / ar_loop / decode), HIP events on the
        current stream (the chain streams are joined before each mark)."""
        marks = []

        def mark(name):
            if timings is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((name, ev))
        mark("start")
        enc = self.encode_cloud(Xct)
        mark("encode")
        g = self.gpt
        B = Xct.shape[0]
        n_micro = n_micro if n_micro is not None else default_chains(B)
        kw = dict(max_steps=max_steps, top_k=top_k, top_p=top_p, temperature=temperature, best_in_first=best_in_first,
                  mask_invalid=mask_invalid, mask_invalid_completion=mask_invalid_completion, seed=seed, stop_early=stop_early)
        if n_micro > 1:
            res = g.sample_microbatched(enc["c_tokens"], enc["Lc"], n_micro=n_micro, after_prefill=lambda: mark("prefill"), **kw)
        else:
            res = g.sample(enc["c_tokens"], enc["Lc"], to_host=False, after_prefill=lambda: mark("prefill"), **kw)
        mark("ar_loop")
        st = res["state"]
        dense = T.sparse2dense_dev(st["seq"], st["len"], enc["empty_index"], self.R, self.end, start=st["Lc"])
        out = self.vq.decode_index(dense, grid_Q=decode_res, sigmoid=sigmoid)
        mark("decode")
        if timings is not None:
            torch.cuda.synchronize()
            for (_, e0), (name, e1) in zip(marks, marks[1:]):
                timings[name] = e0.elapsed_time(e1)
        return dict(c_ind=enc["c_tokens"], Lc=enc["Lc"], empty_index=enc["empty_index"], state=st, steps=res["steps"],
                    dense=dense, occupancy=out["logits"][..., 0], log_prob=st["logp"])

    @torch.no_grad()
    def reconstruct(self, Xbd, decode_res=128, max_length=512, sigmoid=False):
        """VisSparseRecon3D.compute_batch (vqdif.py:243-269): quantize -> sparse -> dense -> decode_index."""
        q, mode, raw, mask, latent = self.vq.quantize_cloud_dev(Xbd, per_shape_mode=False)
        mode2 = T.mode_i32(q, self.vq.K + 1)
        tok, ln = T.dense2sparse_dev(q, mode2, max_length, self.end, Lpad=max_length)
        dense = T.sparse2dense_dev(tok, ln, mode2, self.R, self.end)
        out = self.vq.decode_index(dense, grid_Q=decode_res, sigmoid=sigmoid)
        return dict(logits=out["logits"], quant_ind=raw, sparse=(tok, ln), grid_mask=mask, dense=dense)
