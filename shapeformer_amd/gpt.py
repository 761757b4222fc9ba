"""Host-side mirror of CondTupleGPT + ShapeFormer.sample_indices over libsfmi (HIP, gfx950).

Mirrors shapeformer/models/shapeformer/transformer/mingpt.py:185-319 (state-dict key names, two-stage
tuple head) and shapeformer/models/shapeformer/shapeformer.py:54-130 (`sample_indices` / `sample`),
re-designed as prefill + KV-cached decode: every arithmetic step is a C-ABI call into libsfmi.so
(csrc/gpt.hip, csrc/conv3d.hip:sfmi_gemm_f32); the decode step reads all per-row state from device
memory and is captured once in a hipGraph (torch.cuda.CUDAGraph is only the capture/replay plumbing).

Rows are ragged: each of the B sequences has its own condition length Lc[b]; row b reproduces what the
reference computes for that shape alone at batch size 1 (its inference driver asserts batch_size == 1,
shapeformer.py:227).  No CPU fallback.
"""
from __future__ import annotations

import os
import numpy as np
import torch

from . import _lib as L
from . import weights as W


def _t(sd, k, dev):
    v = sd[k]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(v)
    return v.to(dev, torch.float32).contiguous()


def pack_skinny16(w: torch.Tensor) -> torch.Tensor:
    """(N,K) row-major -> [ceil(N/16)][K/16][64][4] (same as sfmi_skinny16_pack_weight)."""
    N, K = w.shape
    NT = (N + 15) // 16
    if NT * 16 != N:
        w = torch.cat([w, w.new_zeros(NT * 16 - N, K)], 0)
    return w.view(NT, 16, K // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


class _Layer:
    pass


class CondTupleGPT:
    S_PROJ, S_FC2 = 1, 4   # in-kernel split-K of the N = n_embd GEMMs (64 n-tiles -> 256 workgroups)
    S_PROJ_M = 1           # proj above 16 rows: with four chains in flight 1 beats 2 beats 4 (4.13 / 4.17 / 4.18 ms per step); fc2: 4 beats 2 and 8
    PREFILL_ON_CHAIN_STREAMS = True   # tools/probe_chain_streams.py switches it off for the A/B
    # prefill GEMMs with at least this many rows go to the library sgemm; None (default): every GEMM is csrc/sgemm.hip
    PREFILL_BLAS_ROWS = 2048 if os.environ.get("SFMI_ROCBLAS") == "1" else None
    # True: the plain GEMMs of the prefill / teacher-forced forward run on the r1 tile kernel (csrc/conv3d.hip:sfmi_gemm_f32)
    # instead of csrc/sgemm.hip - the alternative in-tree implementation, kept as a cross-check (tests/test_gpt_fullsize_gpu.py)
    PREFILL_TILE_KERNEL = False

    def _blas(self):
        if not hasattr(self, "_has_blas"):
            self._has_blas = bool(L.lib().sfmi_blas_available())
        return self._has_blas

    def __init__(self, state_dict=None, n_embd=1024, n_head=16, n_layers=(20, 4), block_size=812,
                 vocab_sizes=(4097, 4097), extra_vocab_sizes=(4097,), end_tokens=(4096, 4096), device="cuda:0",
                 tuple_n=2, embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, **_ignored):
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise L.SfmiError("CondTupleGPT needs a HIP device (no CPU fallback)")
        L.lib()
        assert tuple_n == 2 and len(n_layers) == 2 and vocab_sizes[0] == vocab_sizes[1]
        self.D, self.H, self.n_layers, self.Lmax = n_embd, n_head, tuple(n_layers), block_size
        self.V, self.end = vocab_sizes[0], tuple(end_tokens)
        self.pdrop = (float(embd_pdrop), float(resid_pdrop), float(attn_pdrop))   # training only (train.GPTTrainer); inference = eval
        self.Vpad = (self.V + 31) // 32 * 32
        sd = state_dict if state_dict is not None else self._hash_state_dict()
        self.load_state_dict(sd)
        self._state = None
        self._states, self._graphs = {}, {}
        # debug / measurement hooks, set explicitly by bench.py and tools/ (never read from the environment on the product path):
        #   _ablate     : TIMING-ONLY ablation of decode_step, "gemm" / "attn" / "gemm@0,attn@1" (per chain): the named kernel
        #                 family is not launched, sampled tokens are garbage; part of the hipGraph cache key
        self._ablate = ""
        #   _profile    : in-situ launch timing of the decode step (csrc/gpt.hip:prof_begin / prof_end_last), "" / "attn" /
        #                 "attn,gemm": the named kernel families add their launch durations to the chain's st["prof"] sink;
        #                 results are untouched; part of the hipGraph cache key.  Read with `launch_profile()`.
        self._profile = ""
        self._sem = torch.zeros(4, device=self.dev, dtype=torch.int32)   # attention turnstile {next ticket, finished, time-outs}

    def _sync_params(self):
        """A trainer that leaves parameter updates in flight (train.GPTTrainer, grad_sync "rs_ag": all-gathers waited for lazily by the next
        training forward) registers its drain here; every OTHER reader of the parameter tensors - sampling, the teacher-forced forward,
        state_dict, the decode-weight refresh - calls it first, whichever entry point it came through."""
        ref = getattr(self, "_param_sync", None)      # weakref.WeakMethod: the model must not keep a dropped trainer (3.9 GB) alive
        hook = ref() if ref is not None else None
        if hook is not None:
            hook()

    def get_block_size(self):
        return self.Lmax

    def _hash_state_dict(self):
        spec = W.gpt_spec(self.D, self.n_layers, self.Lmax, (self.V, self.V), (self.V,))
        return {k: W.make_tensor_torch(k, shp, self.dev) for k, shp in spec.items()}

    # ------------------------------------------------------------------ weights
    def state_dict(self):
        """The reference's 419-tensor key set (mingpt.py:187-254) from the live device parameters: the fused QKV
        rows are split back into query / key / value, the causal-mask buffers are regenerated (tril), tensors on the CPU."""
        self._sync_params()
        D, sd = self.D, {}
        cpu = lambda t: t.detach().cpu().clone()
        sd["pos_emb"], sd["cond_pos_emb"] = cpu(self.pos_emb).view(1, -1, D), cpu(self.cond_pos_emb).view(1, -1, D)
        sd["tok_embs.0.weight"], sd["tok_embs.1.weight"], sd["extra_tok_embs.0.weight"] = cpu(self.E[0]), cpu(self.E[1]), cpu(self.Ex)
        mask = torch.tril(torch.ones(self.Lmax, self.Lmax)).view(1, 1, self.Lmax, self.Lmax)
        li = 0
        for s_, nl in enumerate(self.n_layers):
            for n in range(nl):
                ly, p = self.layers[li], f"blocks.{s_}.{n}."
                li += 1
                sd[p + "ln1.weight"], sd[p + "ln1.bias"] = cpu(ly.ln1[0]), cpu(ly.ln1[1])
                sd[p + "ln2.weight"], sd[p + "ln2.bias"] = cpu(ly.ln2[0]), cpu(ly.ln2[1])
                for i, nm in enumerate(("query", "key", "value")):
                    sd[p + f"attn.{nm}.weight"] = cpu(ly.wqkv[i * D:(i + 1) * D])
                    sd[p + f"attn.{nm}.bias"] = cpu(ly.bqkv[i * D:(i + 1) * D])
                sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = cpu(ly.wproj), cpu(ly.bproj)
                sd[p + "attn.mask"] = mask.clone()
                sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"] = cpu(ly.wfc1), cpu(ly.bfc1)
                sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"] = cpu(ly.wfc2), cpu(ly.bfc2)
        for s_ in range(2):
            sd[f"heads.{s_}.0.weight"], sd[f"heads.{s_}.0.bias"] = cpu(self.head_ln[s_][0]), cpu(self.head_ln[s_][1])
            sd[f"heads.{s_}.1.weight"] = cpu(self.head_w[s_])
        return sd

    def load_state_dict(self, sd):
        dev = self.dev
        g = lambda k: _t(sd, k, dev)
        self.pos_emb = g("pos_emb").view(-1, self.D)
        self.cond_pos_emb = g("cond_pos_emb").view(-1, self.D)
        self.E = [g("tok_embs.0.weight"), g("tok_embs.1.weight")]
        self.Ex = g("extra_tok_embs.0.weight")
        self.layers = []
        for s, nl in enumerate(self.n_layers):
            for n in range(nl):
                p = f"blocks.{s}.{n}."
                ly = _Layer()
                ly.ln1 = (g(p + "ln1.weight"), g(p + "ln1.bias"))
                ly.ln2 = (g(p + "ln2.weight"), g(p + "ln2.bias"))
                # fused QKV rows: [query | key | value] (csrc/gpt.hip attn kernels expect this order)
                ly.wqkv = torch.cat([g(p + "attn.query.weight"), g(p + "attn.key.weight"), g(p + "attn.value.weight")], 0).contiguous()
                ly.bqkv = torch.cat([g(p + "attn.query.bias"), g(p + "attn.key.bias"), g(p + "attn.value.bias")], 0).contiguous()
                ly.wproj, ly.bproj = g(p + "attn.proj.weight"), g(p + "attn.proj.bias")
                ly.wfc1, ly.bfc1 = g(p + "mlp.0.weight"), g(p + "mlp.0.bias")
                ly.wfc2, ly.bfc2 = g(p + "mlp.2.weight"), g(p + "mlp.2.bias")
                ly.stage = s
                self.layers.append(ly)
        self.head_ln = [(g(f"heads.{s}.0.weight"), g(f"heads.{s}.0.bias")) for s in range(2)]
        # head weights live in their GEMM-padded (Vpad, D) buffers; head_w are views of the first V rows, so the optimizer's
        # in-place updates reach the padded copies the teacher-forced forward multiplies with
        self.head_w_pad = []
        for s in range(2):
            w = g(f"heads.{s}.1.weight")
            wp = torch.zeros(self.Vpad, self.D, device=dev)
            wp[:self.V] = w
            self.head_w_pad.append(wp)
        self.head_w = [wp[:self.V] for wp in self.head_w_pad]
        self.zero_bqkv = torch.zeros(3 * self.D, device=dev)
        self.refresh_decode_weights()

    def refresh_decode_weights(self):
        """(Re)build the decode-path weights from the raw parameters: LayerNorm folded into the GEMM
        (csrc/gpt.hip dgemm_kernel):  LN(x) W^T + b = rstd (x W'^T - mean c1) + c2,  W' = W diag(gamma),
        c1 = rowsum(W'), c2 = W beta + b; all matrices in 16x16x4-MFMA fragment order.  Call after the raw
        weights change (training)."""
        self._sync_params()
        def fold(w, bias, ln):      # one in-tree launch per matrix (csrc/gpt.hip:ln_fold_pack_kernel): no library GEMV / reduction at load time
            N, K = w.shape
            Np = (N + 15) // 16 * 16
            wp = torch.empty(Np * K, device=self.dev)
            c1, c2 = (torch.empty(Np, device=self.dev), torch.empty(Np, device=self.dev)) if ln is not None else (None, None)
            L.check(L.lib().sfmi_ln_fold_pack_f32(L.ptr(w), L.ptr(ln[0]) if ln else None, L.ptr(ln[1]) if ln else None, L.ptr(bias),
                                                  L.ptr(wp), L.ptr(c1), L.ptr(c2), N, K, L.stream_ptr()), "sfmi_ln_fold_pack_f32")
            return wp, c1, c2
        for ly in self.layers:
            ly.pqkv, ly.c1qkv, ly.c2qkv = fold(ly.wqkv, ly.bqkv, ly.ln1)
            ly.pfc1, ly.c1fc1, ly.c2fc1 = fold(ly.wfc1, ly.bfc1, ly.ln2)
            ly.pproj, ly.pfc2 = fold(ly.wproj, None, None)[0], fold(ly.wfc2, None, None)[0]
        self.head_f = [fold(self.head_w[s], None, self.head_ln[s]) for s in range(2)]
        self._graphs = {}
        self._decode_stale = False

    def mark_decode_weights_stale(self):
        """The raw parameters changed (optimizer step): re-fold / re-pack lazily, at the next decode use."""
        self._decode_stale = True

    # ------------------------------------------------------------------ state
    def _alloc(self, B, max_steps, slot=0):
        key = (B, max_steps)
        if not hasattr(self, "_states"):
            self._states, self._graphs = {}, {}
        cur = self._states.get(slot)
        if cur is not None and cur["key"] == key:
            self._state = cur
            return cur
        dev, D = self.dev, self.D
        f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        # decode activations are fragment-packed in 16-row tiles (csrc/gpt.hip pk_off); above 96 rows the decode GEMM works on
        # row groups of equal tile counts, so the buffers hold groups x tiles-per-group x 16 rows
        Bp = int(L.lib().sfmi_decode_gemm_padded_rows(B))
        st = dict(key=key,
                  seq=torch.zeros(B, self.Lmax + 1, 2, device=dev, dtype=torch.int32),
                  len=torch.zeros(B, device=dev, dtype=torch.int32), Lc=torch.zeros(B, device=dev, dtype=torch.int32),
                  resid=torch.zeros(Bp, D, device=dev), qkv=torch.zeros(Bp, 3 * D, device=dev), y=torch.zeros(Bp, D, device=dev),
                  h=torch.zeros(Bp, 4 * D, device=dev), logit=f(B, self.Vpad),
                  slab=f(L.lib().sfmi_decode_gemm_slab_floats(Bp, 4 * D, 4)), cnt=torch.zeros(Bp // 16 * (max(4 * D, self.Vpad) // 16 + 1), device=dev, dtype=torch.int32),
                  Kc=f(len(self.layers), B, self.Lmax + 1, D), Vc=f(len(self.layers), B, self.Lmax + 1, D),
                  logp=torch.zeros(B, max_steps, 2, device=dev, dtype=torch.float32),
                  blk=torch.zeros(1, device=dev, dtype=torch.int32),      # attention turnstile: this chain's finished-workgroup counter
                  pblk=torch.zeros(1, device=dev, dtype=torch.int32),     # in-situ launch timing: finished workgroups of the decode-GEMM launch
                  prof=self._prof_zero(dev),                              # in-situ launch timing sinks {t0, sum, launches, -} x {attn, gemm}
                  shared=torch.zeros(1, device=dev, dtype=torch.int32),  # shared-prefix length of the sample_n mode (device-resident)
                  seed=torch.zeros(1, device=dev, dtype=torch.int32))   # sampler seed (device-resident: graphs are seed-independent)
        self._state = st
        self._states[slot] = st
        self._graphs.pop(slot, None)
        return st

    # ------------------------------------------------------------------ in-situ launch timing (bench.py `roofline`)
    @staticmethod
    def _prof_zero(dev):
        t = torch.zeros(8, device=dev, dtype=torch.int64)
        t[0] = -1
        t[4] = -1       # the "earliest start" slots are armed as ~0 (csrc/gpt.hip:prof_end_last re-arms them)
        return t

    def launch_profile(self, reset=True):
        """Launch durations the decode kernels recorded themselves since the last reset (self._profile selects the families):
        {"attn": (launches, mean us), "gemm": (launches, mean us)} summed over all chain states; ticks are 10 ns."""
        out = {}
        sts = [st for st in self._states.values() if "prof" in st]
        for i, fam in enumerate(("attn", "gemm")):
            n = sum(int(st["prof"][4 * i + 2].item()) for st in sts)
            tk = sum(int(st["prof"][4 * i + 1].item()) for st in sts)
            out[fam] = (n, tk * 0.01 / n if n else 0.0)
        if reset:
            for st in sts:
                st["prof"].copy_(self._prof_zero(self.dev))
                st["pblk"].zero_()
        return out

    # ------------------------------------------------------------------ C-ABI wrappers
    def _dgemm(self, x, wp, c1, c2, resid, out, M, N, K, ldo, ln, act, packed=1, S=1, st=None, prof=False):
        st = st or self._state
        while S > 1 and (K // S) % 128:
            S //= 2
        slab, cnt = (L.ptr(st["slab"]), L.ptr(st["cnt"])) if S > 1 else (None, None)
        if prof:
            L.check(L.lib().sfmi_decode_gemm_prof_f32(L.ptr(x), L.ptr(wp), L.ptr(c1), L.ptr(c2), L.ptr(resid), L.ptr(out), M, N, K, ldo,
                                                      ln, act, packed, S, slab, cnt, L.ptr(st["pblk"]), L.ptr(st["prof"][4:]),
                                                      L.stream_ptr()), "sfmi_decode_gemm_prof_f32")
            return
        L.check(L.lib().sfmi_decode_gemm_f32(L.ptr(x), L.ptr(wp), L.ptr(c1), L.ptr(c2), L.ptr(resid), L.ptr(out), M, N, K, ldo,
                                             ln, act, packed, S, slab, cnt, L.stream_ptr()), "sfmi_decode_gemm_f32")

    def _rowprep(self, resid_in, part, bias, S, M, resid_out, xn, ln, Eadd=None, P=0, st=None):
        L.check(L.lib().sfmi_gpt_rowprep_f32(L.ptr(resid_in), L.ptr(part), L.ptr(bias), L.ptr(Eadd),
                                             L.ptr(st["seq"]) if st else None, L.ptr(st["len"]) if st else None,
                                             L.ptr(st["Lc"]) if st else None, L.ptr(st.get("nval")) if st else None,
                                             L.ptr(resid_out), L.ptr(xn),
                                             L.ptr(ln[0]) if ln else None, L.ptr(ln[1]) if ln else None, S, M, P, self.D,
                                             self.Lmax + 1, L.ptr(st.get("rowoff")) if st else None,
                                             st["seq"].shape[0] if st else 0, L.stream_ptr()), "sfmi_gpt_rowprep_f32")

    def _embed(self, st, B, P, resid, xn, ln):
        L.check(L.lib().sfmi_gpt_embed_f32(L.ptr(self.E[0]), L.ptr(self.E[1]), L.ptr(self.Ex), L.ptr(self.pos_emb),
                                           L.ptr(self.cond_pos_emb), L.ptr(st["seq"]), L.ptr(st["len"]), L.ptr(st["Lc"]),
                                           L.ptr(st.get("nval")), L.ptr(st.get("extra")), L.ptr(st.get("extra_out")),
                                           L.ptr(resid), L.ptr(xn), L.ptr(ln[0]) if ln else None, L.ptr(ln[1]) if ln else None,
                                           B, P, self.D, self.Lmax + 1,
                                           self.end[0], L.ptr(st.get("rowoff")), int(st.get("M_packed") or 0), L.stream_ptr()), "sfmi_gpt_embed_f32")

    def _gemm(self, x, w, bias, resid, y, M, N, K, act=0, og=0, ogs=0):
        """y (M,N) = act(x (M,K) w (N,K)^T + bias) + resid: the plain GEMMs of the prefill / teacher-forced forward
        (mingpt.py:46-111 Linear layers).  csrc/sgemm.hip; its per-row result does not depend on how many rows are in the
        launch, so sampled tokens are bit-identical however the rows are split into chains.  PREFILL_BLAS_ROWS (None by
        default; SFMI_ROCBLAS=1 sets 2048) sends long prefills to the library sgemm instead - its kernel choice depends on M,
        so that identity then holds only to fp32 rounding."""
        calls = self.__dict__.setdefault("_gemm_calls", {"blas": 0, "sgemm": 0, "tile": 0})   # which kernel ran (tests / diagnostics)
        if self.PREFILL_BLAS_ROWS is not None and M >= self.PREFILL_BLAS_ROWS and not og and N % 4 == 0 and self._blas() \
                and not (resid is not None and resid.data_ptr() == y.data_ptr() and act):
            calls["blas"] += 1
            L.check(L.lib().sfmi_gemm_blas_f32(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(resid), L.ptr(y), M, N, K, act, L.stream_ptr()), "gemm_blas")
            return
        if not og and N % 4 == 0 and K % 4 == 0 and not self.PREFILL_TILE_KERNEL:
            calls["sgemm"] += 1
            L.check(L.lib().sfmi_sgemm_mfma_f32(0, 1, M, N, K, L.ptr(x), K, L.ptr(w), K, L.ptr(y), N, 0, L.ptr(bias), act, L.ptr(resid),
                                                None, 0, 0.0, 0, L.stream_ptr()), "sfmi_sgemm_mfma_f32")
            return
        calls["tile"] += 1
        L.check(L.lib().sfmi_gemm_f32(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(resid), L.ptr(y), M, N, K, act, og, ogs,
                                      L.stream_ptr()), "sfmi_gemm_f32")

    # ------------------------------------------------------------------ prefill (positions 0..nval[b]-1)
    def prefill(self, st, B, P, want_logits=False):
        """Batched forward of rows (b,t), t < P, through both stages (fills the KV caches).  Rows with
        t >= nval[b] (default Lc[b]-1) are padding.  want_logits: also run both heads on every row and return
        [(B,P,V), (B,P,V)] — the teacher-forced `CondTupleGPT.forward` (mingpt.py:287-296,311-319)."""
        gen = self._prefill_gen(st, B, P, want_logits, interactive=False)
        try:
            while True:
                next(gen)
        except StopIteration as e:
            return e.value

    def _prefill_gen(self, st, B, P, want_logits, interactive):
        """Generator form of `prefill`: with interactive=True it yields the stage-0 logits, expects the chosen
        positions (B,P) via .send() (they become seq[:, 1:P+1, 0], the tok_embs[0] add of mingpt.py:309), then yields the
        stage-1 logits — the protocol of CondTupleGPT.sample_next_tuple (mingpt.py:297-310)."""
        D, dev = self.D, self.dev
        rowoff = st.get("rowoff")                 # packed ragged rows (sampling prefill) or the (B,P) rectangle
        M = int(st["M_packed"]) if rowoff is not None else B * P
        assert rowoff is None or not want_logits
        f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        resid, xn, qkv, y, h = f(M, D), f(M, D), f(M, 3 * D), f(M, D), f(M, 4 * D)
        if st.get("nval") is None:
            st["nval"] = (st["Lc"] - 1).clamp_(min=0).contiguous()
        logits = []

        def heads(s):
            if not want_logits:
                return
            self._rowprep(resid, None, None, 0, M, None, xn, self.head_ln[s])
            lg = f(M, self.Vpad)
            self._gemm(xn, self.head_w_pad[s], None, None, lg, M, self.Vpad, D)
            logits.append(lg.view(B, P, self.Vpad)[:, :, :self.V])

        self._embed(st, B, P, resid, xn, self.layers[0].ln1)
        for li, ly in enumerate(self.layers):
            self._gemm(xn, ly.wqkv, ly.bqkv, None, qkv, M, 3 * D, D)
            L.check(L.lib().sfmi_gpt_attn_prefill_f32(L.ptr(qkv), L.ptr(st["Kc"][li]), L.ptr(st["Vc"][li]), L.ptr(st["nval"]),
                                                      L.ptr(y), B, P, D, self.H, self.Lmax + 1, L.ptr(rowoff), 0.0, 0, L.stream_ptr()), "attn_prefill")
            self._gemm(y, ly.wproj, ly.bproj, resid, resid, M, D, D)
            self._rowprep(resid, None, None, 0, M, None, xn, ly.ln2)
            self._gemm(xn, ly.wfc1, ly.bfc1, None, h, M, 4 * D, D, act=2)
            self._gemm(h, ly.wfc2, ly.bfc2, resid, resid, M, D, 4 * D)
            nxt = self.layers[li + 1] if li + 1 < len(self.layers) else None
            if nxt is None:
                heads(ly.stage)
                if interactive:
                    yield logits[-1]
                break
            if nxt.stage != ly.stage:   # stage boundary: x += tok_embs[0][next pos] (mingpt.py:294)
                heads(ly.stage)
                if interactive:
                    tgt = yield logits[-1]
                    if tgt is not None:
                        st["seq"][:, 1:P + 1, 0] = torch.as_tensor(tgt).to(self.dev, torch.int32)
                self._rowprep(resid, None, None, 0, M, resid, xn, nxt.ln1, Eadd=self.E[0], P=P, st=st)
            else:
                self._rowprep(resid, None, None, 0, M, None, xn, nxt.ln1)
        return logits

    # ------------------------------------------------------------------ teacher-forced forward (mingpt.py:311-319)
    @torch.no_grad()
    def forward(self, idx, extra_idx=None, L_cond=1, target_idx=None):
        """CondTupleGPT.forward: idx (B,L,2), extra_idx (B,L,1) or None (-> AR_N rule), target_idx (B,L,2)
        -> [logits_pos (B,L,V), logits_val (B,L,V)] (float32, on device)."""
        self._sync_params()
        idx = torch.as_tensor(idx).to(self.dev, torch.int32)
        B, Lq, _ = idx.shape
        assert Lq <= self.Lmax, "Cannot forward, model block size is exhausted."   # mingpt.py:279
        st = dict(self._alloc(B, 1))
        st["seq"] = torch.zeros(B, self.Lmax + 1, 2, device=self.dev, dtype=torch.int32)
        st["seq"][:, :Lq] = idx
        if target_idx is not None:       # stage-1 input adds tok_embs[0][target pos] = seq[t+1][0]
            tgt = torch.as_tensor(target_idx).to(self.dev, torch.int32)
            assert bool((tgt[:, :-1, 0] == idx[:, 1:, 0]).all()), "target_idx must be idx shifted left (shapeformer.py:37-41)"
            st["seq"][:, Lq, 0] = tgt[:, -1, 0]
        st["Lc"] = torch.full((B,), int(L_cond), device=self.dev, dtype=torch.int32)
        st["len"] = torch.full((B,), Lq, device=self.dev, dtype=torch.int32)
        st["nval"] = torch.full((B,), Lq, device=self.dev, dtype=torch.int32)
        st["extra"] = None
        st["rowoff"], st["M_packed"] = None, 0      # teacher-forced rows are a full (B,L) rectangle
        if extra_idx is not None:
            ex = torch.zeros(B, self.Lmax + 1, device=self.dev, dtype=torch.int32)
            ex[:, :Lq] = torch.as_tensor(extra_idx).to(self.dev, torch.int32)[..., 0]
            st["extra"] = ex
        return self.prefill(st, B, Lq, want_logits=True)

    __call__ = forward

    def _forward_state(self, idx, extra_idx, L_cond):
        self._sync_params()
        idx = torch.as_tensor(idx).to(self.dev, torch.int32)
        B, Lq, _ = idx.shape
        assert Lq <= self.Lmax, "Cannot forward, model block size is exhausted."   # mingpt.py:279
        st = dict(self._alloc(B, 1))
        st["seq"] = torch.zeros(B, self.Lmax + 1, 2, device=self.dev, dtype=torch.int32)
        st["seq"][:, :Lq] = idx
        st["Lc"] = torch.full((B,), int(L_cond), device=self.dev, dtype=torch.int32)
        st["len"] = torch.full((B,), Lq, device=self.dev, dtype=torch.int32)
        st["nval"] = torch.full((B,), Lq, device=self.dev, dtype=torch.int32)
        st["extra"] = None
        st["rowoff"], st["M_packed"] = None, 0      # teacher-forced rows are a full (B,L) rectangle
        if extra_idx is not None:
            ex = torch.zeros(B, self.Lmax + 1, device=self.dev, dtype=torch.int32)
            ex[:, :Lq] = torch.as_tensor(extra_idx).to(self.dev, torch.int32)[..., 0]
            st["extra"] = ex
        return st, B, Lq

    @torch.no_grad()
    def sample_next_tuple(self, idx, extra_idx=None, L_cond=1):
        """Compat generator with the reference protocol (mingpt.py:297-310): `g = m.sample_next_tuple(idx, extra, L_c)`;
        `logits_pos = next(g)` (B,L,V); `logits_val = g.send(target_pos)` with target_pos (B,L) = idx positions shifted
        left + the newly chosen one.  It recomputes the whole prefix like the reference does (use `sample()` for the
        KV-cached device loop); provided so ShapeFormer.sample_indices (shapeformer.py:54-123) can run unchanged."""
        st, B, Lq = self._forward_state(idx, extra_idx, L_cond)
        gen = self._prefill_gen(st, B, Lq, True, interactive=True)
        target = yield next(gen)
        yield gen.send(target)

    @torch.no_grad()
    def training_loss(self, c_indices, z_indices, extra_indices=None):
        """ShapeFormer.forward + shared_step loss (shapeformer.py:26-46,132-140): mean of the two cross entropies over
        the outputs from index L_c-1 on (end-token padding included).  Forward only (no backward in round 1)."""
        c = torch.as_tensor(c_indices).to(self.dev, torch.int32)
        z = torch.as_tensor(z_indices).to(self.dev, torch.int32)
        cz = torch.cat([c, z], 1)
        L_c = c.shape[1]
        ex = None if extra_indices is None else torch.as_tensor(extra_indices)[:, :-1]
        logits = self.forward(cz[:, :-1], ex, L_c, cz[:, 1:])
        loss = 0.0
        for i in range(2):
            lg = logits[i][:, L_c - 1:, :].contiguous()
            M = lg.shape[0] * lg.shape[1]
            tgt = z[..., i].contiguous().view(-1)
            rows = torch.empty(M, device=self.dev, dtype=torch.float32)
            L.check(L.lib().sfmi_ce_rows_f32(L.ptr(lg), L.ptr(tgt), L.ptr(rows), M, self.V, self.V, L.stream_ptr()), "sfmi_ce_rows_f32")
            loss = loss + rows.mean()
        return loss / 2

    # ------------------------------------------------------------------ one decode step (graph-capturable)
    # Attention turnstile of the interleaved decode chains (csrc/gpt.hip:attn_gate_kernel): > 0 = at most this many chains stream
    # their KV cache at the same time (FIFO tickets in device memory); the other chains' GEMM workgroups then always find room
    # on every CU.  0 = off (independent chains, round 2 behaviour).  Scheduling only: tokens / logits are bit-identical.
    # Two lanes measured best with four chains (6.48 against 6.68 ms per 320-row step ungated, 6.98 with one lane;
    # profiles/r03_ar_overlap.md).
    ATTN_LANES = 2
    MAX_CHAIN_ROWS = 192   # rows per decode chain: row groups of up to 6 row tiles (96 rows) per decode-GEMM workgroup; larger batches = several chains
    # shared_prefix="auto" (the sample_n copies of one condition, shapeformer.py:222-260): one prefill and one copy of the condition's
    # keys / values for all rows, bit-identical to expanded rows, but its decode attention walks two caches per row.  Whole-call
    # times of 512 steps (tools/bench_shared_prefix.py, round 5): 16 rows: expanded wins below L_c = 175 (1.020 vs 1.040 ms/step at
    # L_c = 84), shared above (1.153 vs 1.116 at 300); 64 rows: shared wins from L_c = 84 on (2.037 vs 1.965; 2.569 vs 2.168 at 300).
    SHARED_PREFIX_MIN_ROW_TOKENS = 2800      # rows x condition length from which the shared form is taken
    SINGLE_CHAIN_ROWS = 96  # `sample` keeps a batch in ONE chain up to here and interleaves chains above (a lone chain cannot overlap anything)

    def decode_step(self, st, B, sp):
        """Position t = len[b]-1 of every row through both stages; st["resid"] must hold its embedding on entry
        (written by the previous step's sampler tail, or by `_embed` before the first step)."""
        D = self.D
        lib = L.lib()
        r = st["resid"]
        skip = self._ablate                               # timing-only ablation hook (see __init__): results are garbage
        if "@" in skip:      # per-chain form "gemm@0,attn@1,attn@2": chain index = micro-batch slot
            skip = ",".join(t.split("@")[0] for t in skip.split(",") if int(t.split("@")[1]) == sp.get("chain", 0))
        lanes = int(sp.get("gate_lanes", 0))
        pa, pg = "attn" in self._profile, "gemm" in self._profile      # in-situ launch timing (results untouched)
        # in-kernel split-K per GEMM (only the K = 4 n_embd product, and proj at <= 16 rows, use it)
        Sqkv, Sproj, Sfc1, Sfc2, Shead = 1, self.S_PROJ if B <= 16 else self.S_PROJ_M, 1, self.S_FC2, 1
        for li, ly in enumerate(self.layers):
            stage_end = li + 1 == len(self.layers) or self.layers[li + 1].stage != ly.stage
            if "gemm" not in skip:
                self._dgemm(r, ly.pqkv, ly.c1qkv, ly.c2qkv, None, st["qkv"], B, 3 * D, D, 3 * D, 1, 0, S=Sqkv, st=st, prof=pg)
            if "attn" not in skip:
                L.check(lib.sfmi_gpt_attn_decode_gated_f32(L.ptr(st["qkv"]), L.ptr(st["Kc"][li]), L.ptr(st["Vc"][li]),
                                                           L.ptr(st["len"]), L.ptr(st["y"]), B, D, self.H, self.Lmax + 1,
                                                           L.ptr(st["shared"]) if sp.get("shared_prefix") else None,
                                                           L.ptr(self._sem) if lanes else None, L.ptr(st["blk"]) if (lanes or pa) else None, lanes,
                                                           L.ptr(st["prof"]) if pa else None, L.stream_ptr()), "sfmi_gpt_attn_decode_gated_f32")
            if "gemm" not in skip:
                self._dgemm(st["y"], ly.pproj, None, ly.bproj, r, r, B, D, D, D, 0, 0, S=Sproj, st=st, prof=pg)
                self._dgemm(r, ly.pfc1, ly.c1fc1, ly.c2fc1, None, st["h"], B, 4 * D, D, 4 * D, 1, 1, S=Sfc1, st=st, prof=pg)
                self._dgemm(st["h"], ly.pfc2, None, ly.bfc2, r, r, B, D, 4 * D, D, 0, 0, S=Sfc2, st=st, prof=pg)
            if stage_end:
                s = ly.stage
                hp, hc1, hc2 = self.head_f[s]
                self._dgemm(r, hp, hc1, hc2, None, st["logit"], B, self.V, D, self.Vpad, 1, 0, packed=0, S=Shead, st=st)
                hist = sp["hist"][s] if sp.get("hist") is not None else None
                L.check(lib.sfmi_gpt_sample_f32(L.ptr(st["logit"]), L.ptr(st["seq"]), L.ptr(st["len"]), L.ptr(st["Lc"]),
                                                L.ptr(st["logp"]), L.ptr(hist), L.ptr(sp.get("force")),
                                                L.ptr(r), L.ptr(self.E[0]), L.ptr(self.E[1]), L.ptr(self.Ex), L.ptr(self.pos_emb), D,
                                                1, B, self.V, self.Vpad, self.Lmax + 1,
                                                s, self.end[0], self.end[1], sp["top_k"], sp["top_p"], sp["temperature"],
                                                int(sp["best_in_first"]), int(sp["mask_invalid"]),
                                                int(sp["mask_invalid_completion"]), sp["max_steps"], sp["seed"], L.ptr(st.get("seed")), int(s == 1),
                                                sp.get("row_offset", 0), sp.get("rows_total", B), int(sp.get("step_offset", 0)),
                                                L.stream_ptr()), "sfmi_gpt_sample_f32")

    # ------------------------------------------------------------------ sample_indices
    def _prepare(self, c_tokens, Lc, max_steps, sp_kw, slot=0, row_offset=0, rows_total=None, return_logits=False,
                 force_tokens=None, use_graph=True, shared_prefix=False, z_tokens=None, gate_lanes=0):
        """State + prefill + step-0 embedding + (cached) hipGraph of one decode step for one (micro-)batch.
        z_tokens (B,L_z,2): tokens already generated after the condition (shapeformer.py:60-70 copies cat(c, z) into `sampled`):
        they are prefilled together with the condition and sampling continues after them; the step counter restarts at 0."""
        B = c_tokens.shape[0]
        Lz = 0 if z_tokens is None else int(z_tokens.shape[1])
        if B > self.MAX_CHAIN_ROWS:
            raise L.SfmiError(f"a decode chain holds up to {self.MAX_CHAIN_ROWS} rows (larger batches run as several chains: sample / sample_microbatched)")
        self._sync_params()
        if getattr(self, "_decode_stale", False):
            self.refresh_decode_weights()
        Lc_host = Lc.cpu().tolist()
        Lc_max = max(Lc_host)
        if shared_prefix == "auto":      # the caller vouches for identical condition rows; shared only where it pays
            shared_prefix = B * Lc_max >= self.SHARED_PREFIX_MIN_ROW_TOKENS
        steps = min(max_steps, self.Lmax - Lc_max - Lz)   # never exceed block_size (stop, don't crop)
        st = self._alloc(B, max_steps, slot)
        st["seq"].zero_()
        st["seq"][:, :c_tokens.shape[1]] = c_tokens.to(self.dev, torch.int32)
        st["Lc"].copy_(Lc.to(self.dev, torch.int32))
        st["len"].copy_(st["Lc"])
        if Lz:      # row b: positions Lc[b] .. Lc[b]+Lz-1 hold its z tokens (rows are ragged in Lc)
            pos = st["Lc"].long()[:, None] + torch.arange(Lz, device=self.dev)[None, :]
            st["seq"][torch.arange(B, device=self.dev)[:, None], pos] = torch.as_tensor(z_tokens).to(self.dev, torch.int32)
            st["len"].add_(Lz)
            shared_prefix = False
        st["logp"].zero_()
        st["seed"].copy_(torch.from_numpy(np.array([sp_kw["seed"]], np.uint32).view(np.int32)))
        hist = None
        if return_logits:
            hist = [torch.full((B, max_steps, self.V), float("nan"), device=self.dev) for _ in range(2)]
        sp = dict(sp_kw, max_steps=int(max_steps), hist=hist, row_offset=int(row_offset),
                  rows_total=int(rows_total if rows_total is not None else B), chain=int(slot - 100 if slot >= 100 else 0),
                  shared_prefix=bool(shared_prefix), step_offset=Lz, gate_lanes=int(gate_lanes))
        if force_tokens is not None:   # (B,max_steps,2) teacher forcing for stepwise parity tests
            ft = torch.zeros(B, max_steps, 2, dtype=torch.int32)
            ft[:, :force_tokens.shape[1]] = torch.as_tensor(force_tokens).to(torch.int32)
            sp["force"] = ft.to(self.dev)
            use_graph = False
        P = Lc_max + Lz - 1
        st["nval"], st["extra"] = None, None
        if Lz:
            st["nval"] = (st["len"] - 1).contiguous()     # prefill rows t < Lc + Lz - 1; the last z token is step 0's input
        if shared_prefix:
            # all rows carry the SAME condition (the sample_n copies of one shape, shapeformer.py:222-260): prefill it ONCE,
            # as row 0; its keys / values (positions < Lc-1) are then read from row 0's cache by every row's decode attention
            assert len(set(Lc_host)) == 1 and bool((st["seq"][:, :Lc_max] == st["seq"][:1, :Lc_max]).all()), \
                "shared_prefix needs identical condition rows"
            st["shared"].fill_(P)
            st["M_packed"] = P
            st["rowoff"] = torch.tensor([0, P], dtype=torch.int32).to(self.dev)
            if P > 0:
                self.prefill(st, 1, P)
        else:
            st["shared"].zero_()
            # ragged condition prefixes are packed back to back: the prefill GEMMs / attention do no work on padding rows
            nrow = [max(l + Lz - 1, 0) for l in Lc_host]
            st["M_packed"] = sum(nrow)
            st["rowoff"] = torch.tensor([0] + list(np.cumsum(nrow)), dtype=torch.int32).to(self.dev)
            if P > 0 and st["M_packed"] > 0:
                self.prefill(st, B, P)
        # embedding of the last condition token (step-0 input) into the fragment-packed residual buffer
        L.check(L.lib().sfmi_gpt_embed_packed_f32(L.ptr(self.E[0]), L.ptr(self.E[1]), L.ptr(self.Ex), L.ptr(self.pos_emb),
                                                  L.ptr(self.cond_pos_emb), L.ptr(st["seq"]), L.ptr(st["len"]), L.ptr(st["Lc"]),
                                                  L.ptr(st["resid"]), B, self.D, self.Lmax + 1, self.end[0], L.stream_ptr()),
                "sfmi_gpt_embed_packed_f32")
        graph = None
        if use_graph and steps > 1:
            gkey = (B, tuple(sorted((k, v) for k, v in sp.items() if k not in ("hist", "force", "seed"))), return_logits,
                    self._ablate, self._profile, self.S_PROJ, self.S_PROJ_M, self.S_FC2,
                    int(L.lib().sfmi_tune_generation()))      # launch-shape knobs are baked into a captured graph: re-capture when one changed
            cached = self._graphs.get(slot)
            if cached is None or cached[0] != gkey or return_logits:
                side = torch.cuda.Stream(device=self.dev)
                side.wait_stream(torch.cuda.current_stream())
                saved = {k: st[k].clone() for k in ("seq", "len", "logp", "resid")}
                with torch.cuda.stream(side):
                    self.decode_step(st, B, sp)      # warm-up outside capture
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                for k, v in saved.items():
                    st[k].copy_(v)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self.decode_step(st, B, sp)
                for k, v in saved.items():           # capture does not execute, but keep state pristine
                    st[k].copy_(v)
                self._graphs[slot] = (gkey, graph)
            graph = self._graphs[slot][1]
        return dict(st=st, sp=sp, B=B, steps=steps, graph=graph, hist=hist, Lc_host=Lc_host, Lz=Lz)

    @staticmethod
    def _sp(top_k, top_p, temperature, best_in_first, mask_invalid, mask_invalid_completion, seed):
        return dict(top_k=int(top_k), top_p=float(top_p), temperature=float(temperature), best_in_first=best_in_first,
                    mask_invalid=mask_invalid, mask_invalid_completion=mask_invalid_completion,
                    seed=W._fnv1a32(f"sample-uniforms-{seed}"))

    @torch.no_grad()
    def sample(self, c_tokens, Lc, max_steps=512, top_k=100, top_p=0.4, temperature=1.0, best_in_first=True,
               mask_invalid=True, mask_invalid_completion=True, seed=0, stop_early=True, use_graph=True,
               return_logits=False, check_every=32, force_tokens=None, to_host=True, after_prefill=None, shared_prefix=False,
               z_tokens=None, row_offset=0, rows_total=None, ended_reduce=None):
        """c_tokens (B,Lpad,2) int32 (row b valid for Lc[b] tokens, last = end-token pair), Lc (B,) int32; z_tokens (B,L_z,2):
        optional tokens already generated (sampling continues after them; `samples` then starts with them, as the reference's
        x = sampled[:, L_c:] does, shapeformer.py:121, while log_prob / logits_history cover the NEW steps only).

        row_offset / rows_total: these B rows are rows [row_offset, row_offset + B) of a batch of rows_total (another process holds
        the others, dist.sample_n_sharded): the uniform stream and the greedy row are indexed by the GLOBAL row, so the tokens
        are those of the whole batch in one process.  ended_reduce(bool) -> bool: combines this process's "all my rows have ended"
        with the other processes' at every early-stop check (the reference stops when ALL rows have ended, shapeformer.py:110-115).

        Returns dict(samples (B,L_z+steps,2) int64, log_prob (B,steps,2), steps, [logits_history]).
        Mirrors ShapeFormer.sample_indices (shapeformer.py:54-123); torch.multinomial is replaced by an
        inverse-CDF draw on counter-hash uniforms (oracle/gpt_oracle.py:uniforms)."""
        if c_tokens.shape[0] > self.SINGLE_CHAIN_ROWS:
            # more rows than one chain should hold: the same rows as interleaved chains (identical tokens - uniforms and the
            # greedy row are indexed by global row), results gathered as for one chain
            n_micro = min(4, -(-c_tokens.shape[0] // 80))      # up to 4 chains; a chain holds up to MAX_CHAIN_ROWS (192) rows, so one round
                                                               # takes 768 rows (measured best: 4 x 96); beyond that: successive rounds
            r = self.sample_microbatched(c_tokens, Lc, n_micro=n_micro, max_steps=max_steps, top_k=top_k, top_p=top_p, temperature=temperature,
                                         best_in_first=best_in_first, mask_invalid=mask_invalid, mask_invalid_completion=mask_invalid_completion,
                                         seed=seed, stop_early=stop_early, check_every=check_every, after_prefill=after_prefill,
                                         return_logits=return_logits, force_tokens=force_tokens, shared_prefix=shared_prefix,
                                         z_tokens=z_tokens, use_graph=use_graph, _row0=int(row_offset), _rows_total=rows_total,
                                         ended_reduce=ended_reduce)
            if not to_host:
                return r
            Lz = 0 if z_tokens is None else int(z_tokens.shape[1])
            return self._host_result(r["state"], Lc.cpu().tolist(), Lz, r.get("logits_history"))
        sp_kw = self._sp(top_k, top_p, temperature, best_in_first, mask_invalid, mask_invalid_completion, seed)
        ctx = self._prepare(c_tokens, Lc, max_steps, sp_kw, return_logits=return_logits, force_tokens=force_tokens,
                            use_graph=use_graph, shared_prefix=shared_prefix, z_tokens=z_tokens, row_offset=int(row_offset),
                            rows_total=rows_total)
        ended = (lambda: self._all_ended(st, B)) if ended_reduce is None else (lambda: bool(ended_reduce(self._all_ended(st, B))))
        st, sp, B, steps, g, hist, Lc_host = (ctx[k] for k in ("st", "sp", "B", "steps", "graph", "hist", "Lc_host"))
        if after_prefill is not None:
            after_prefill()
        done = 0
        if g is not None:
            while done < steps:
                n = min(check_every, steps - done) if stop_early else steps - done
                for _ in range(n):
                    g.replay()
                done += n
                if stop_early and ended():
                    break
        else:
            while done < steps:
                self.decode_step(st, B, sp)
                done += 1
                # the same check points as the graph loop and sample_microbatched: every check_every steps AND after the last step (a
                # caller's ended_reduce is a collective: every process must reach it the same number of times whatever loop it runs)
                if stop_early and (done % check_every == 0 or done == steps) and ended():
                    break
        if not to_host:   # device-resident result for the completion pipeline (no D2H of tokens)
            return dict(state=st, steps=done)
        return self._host_result(st, Lc_host, ctx["Lz"], hist)

    def _host_result(self, st, Lc_host, Lz, hist):
        """Device state -> the host-side result dict of `sample` (tokens after each row's condition, the z prefix included)."""
        B = len(Lc_host)
        ln = st["len"].cpu().tolist()
        nsteps = max(l - c for l, c in zip(ln, Lc_host))
        out = torch.full((B, nsteps, 2), 0, dtype=torch.int64)
        seq = st["seq"].cpu()
        for b in range(B):
            out[b] = seq[b, Lc_host[b]:Lc_host[b] + nsteps].long()
        n_new = nsteps - Lz
        res = dict(samples=out, log_prob=st["logp"][:, :n_new].cpu(), steps=n_new)
        if hist is not None:
            res["logits_history"] = [h[:, :n_new].cpu() for h in hist]
        return res

    @torch.no_grad()
    def _chain_streams(self, n):
        """n HIP streams that really run concurrently, for the interleaved decode chains.  The runtime multiplexes streams
        onto a few hardware queues (assigned at a stream's first use, least-loaded queue first), and two chains whose streams
        share a queue run back to back instead of overlapping (measured: 5.2 instead of 4.2 ms per decode step when two of the
        three chains share one).  Which streams collide depends on every stream the process used before, so candidates are
        probed: two 200 us single-wavefront spins (csrc/capi.hip) take ~200 us on distinct queues, ~400 us on a shared one."""
        ticks = 20000                       # x 10 ns
        self._chain_probe = getattr(self, "_chain_probe", [])   # probe times in ms, kept for diagnostics
        spin = lambda s, t: L.check(L.lib().sfmi_stream_spin(t, s.cuda_stream), "sfmi_stream_spin")
        cur = torch.cuda.current_stream()

        def overlap(*ss):
            """All of `ss` spin at once: ~0.2 ms when every stream has a hardware queue of its own, >= 0.4 ms when two share one."""
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for s in ss:
                s.wait_event(e0)
                spin(s, ticks)
            for s in ss:
                cur.wait_stream(s)
            e1.record(cur)
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            self._chain_probe.append(round(ms, 3))
            del self._chain_probe[:-32]          # diagnostics only: keep the last 32 probe times (a long-lived service probes forever)
            return ms < 1.5 * ticks * 1e-5

        chosen = list(getattr(self, "_mb_streams", []))
        if len(chosen) >= n and getattr(self, "_mb_shared_queue", False):
            # the last probe found no set of distinct queues: try again from scratch every 64th use (queues free up when other
            # streams of the process die) instead of staying ungated for the life of the process
            self._mb_uses = getattr(self, "_mb_uses", 0) + 1
            if self._mb_uses % 64 == 0:
                chosen, self._mb_streams, self._mb_shared_queue = [], [], False
        if len(chosen) >= n:
            # the cached set is re-checked at every use (one joint 0.2 ms spin): the stream -> queue binding has been seen to change
            # while the streams sat idle between batches (a set that passed the pairwise probe measured 0.43 ms for one pair a few
            # launches later, tests/test_gpt_gpu.py) - two chains on one queue cost 25 % of the loop and stall the turnstile
            if n < 2 or getattr(self, "_mb_shared_queue", False) or overlap(*chosen[:n]) or overlap(*chosen[:n]):
                return chosen[:n]
            chosen = []                     # re-probe from scratch with fresh streams
            self._mb_streams = []
            self._chain_reprobes = getattr(self, "_chain_reprobes", 0) + 1
        spare = []
        for _ in range(16):
            if len(chosen) >= n:
                break
            c = torch.cuda.Stream(device=self.dev)
            spin(c, 1)                       # first use: the runtime binds the stream to a hardware queue here
            c.synchronize()
            if all(overlap(c, x) for x in chosen):
                chosen.append(c)
            else:
                spare.append(c)
        if len(chosen) < n:
            import warnings
            warnings.warn(f"only {len(chosen)} of {n} decode chains get a hardware queue of their own; the others share one")
            chosen += spare[:n - len(chosen)]
            self._mb_shared_queue = True
        else:
            self._mb_shared_queue = False    # a successful (re-)probe re-enables the turnstile
        self._mb_streams = chosen
        return chosen[:n]

    def sample_microbatched(self, c_tokens, Lc, n_micro=2, max_steps=512, top_k=100, top_p=0.4, temperature=1.0,
                            best_in_first=True, mask_invalid=True, mask_invalid_completion=True, seed=0, stop_early=True,
                            check_every=32, after_prefill=None, return_logits=False, force_tokens=None, shared_prefix=False,
                            z_tokens=None, use_graph=True, _row0=0, _rows_total=None, ended_reduce=None):
        """Same result as `sample(..., to_host=False)` (identical tokens: the uniform stream and the greedy row are
        indexed by GLOBAL row), but the rows are split into `n_micro` independent micro-batches whose decode steps are
        separate hipGraphs replayed on separate HIP streams: one micro-batch's HBM-bound attention overlaps the other's
        MFMA-bound GEMMs (each chain is serial, the hardware interleaves the two)."""
        B = c_tokens.shape[0]
        rows_total = B if _rows_total is None else _rows_total
        cap = n_micro * self.MAX_CHAIN_ROWS
        if B > cap:      # more rows than n_micro chains hold: successive rounds of the same shape (rows keep their global index)
            nr = -(-B // cap)
            rb = [round(i * B / nr) for i in range(nr + 1)]
            sl = lambda t, lo, hi: None if t is None else torch.as_tensor(t)[lo:hi]
            def round_(lo, hi, steps, early):
                return self.sample_microbatched(c_tokens[lo:hi], Lc[lo:hi], n_micro=n_micro, max_steps=steps, top_k=top_k, top_p=top_p,
                                                temperature=temperature, best_in_first=best_in_first, mask_invalid=mask_invalid,
                                                mask_invalid_completion=mask_invalid_completion, seed=seed, stop_early=early,
                                                check_every=check_every, after_prefill=after_prefill if lo == 0 else None,
                                                return_logits=return_logits, force_tokens=sl(force_tokens, lo, hi), shared_prefix=shared_prefix,
                                                z_tokens=sl(z_tokens, lo, hi), use_graph=use_graph, _row0=_row0 + lo, _rows_total=rows_total)
            spans = list(zip(rb[:-1], rb[1:]))
            parts = [round_(lo, hi, max_steps, stop_early) for lo, hi in spans]
            # the rounds stop early independently; a single run (shapeformer.py:110-115) keeps stepping every row until ALL rows have
            # ended.  A row whose last POSITION is the end token can only draw end-token pairs from then on (the position masks leave
            # nothing else, log-probability 0): a round that stopped sooner and holds only such rows is padded with exactly those
            # tokens.  A row that "ended" by drawing the end VALUE at a real position is not forced to end tokens (and nothing is with
            # mask_invalid off): such a round is re-run for the longest round's step count without the early stop - rows keep their
            # global index and uniforms, so this is what the single run draws for them.
            nmax = max(p["steps"] for p in parts)
            for i, (lo, hi) in enumerate(spans):
                p_ = parts[i]
                short = nmax - p_["steps"]
                if short <= 0:
                    continue
                stp = p_["state"]
                last = stp["seq"][torch.arange(stp["len"].shape[0], device=self.dev), (stp["len"].long() - 1).clamp(min=0), 0]
                full = stp["len"] >= self.Lmax           # the block is full: nothing more is drawn for this row in a single run either
                if mask_invalid and bool(((last == self.end[0]) | full).all()):
                    room = (self.Lmax - stp["len"]).clamp(min=0, max=short).long()
                    pos = stp["len"].long()[:, None] + torch.arange(short, device=self.dev)[None, :]
                    ok = torch.arange(short, device=self.dev)[None, :] < room[:, None]
                    rows = torch.arange(stp["len"].shape[0], device=self.dev)[:, None].expand_as(pos)
                    stp["seq"][rows[ok], pos[ok]] = torch.tensor(self.end, device=self.dev, dtype=torch.int32)
                    stp["len"] = stp["len"] + room.to(torch.int32)
                    if return_logits:      # the padded steps drew from a one-candidate distribution; their logits rows stay zero
                        p_["logits_history"] = [torch.cat([h, h.new_zeros(h.shape[0], short, h.shape[2])], 1) for h in p_["logits_history"]]
                else:
                    parts[i] = round_(lo, hi, nmax, False)
            res = dict(state={k: torch.cat([p["state"][k] for p in parts], 0) for k in parts[0]["state"]}, steps=nmax)
            if return_logits:
                res["logits_history"] = [torch.cat([p["logits_history"][i] for p in parts], 0) for i in range(2)]
            return res
        bounds = [round(i * B / n_micro) for i in range(n_micro + 1)]
        groups = [(bounds[i], bounds[i + 1]) for i in range(n_micro) if bounds[i + 1] > bounds[i]]
        sp_kw = self._sp(top_k, top_p, temperature, best_in_first, mask_invalid, mask_invalid_completion, seed)
        streams = self._chain_streams(len(groups))
        cur = torch.cuda.current_stream()
        # attention turnstile (ATTN_LANES): only when every chain owns a hardware queue - a gate kernel spinning at the head of a
        # shared queue would hold up the very launch it waits for (it gives up after 20 ms, but that is no way to run)
        lanes = self.ATTN_LANES if (self.ATTN_LANES > 0 and len(groups) > self.ATTN_LANES and not getattr(self, "_mb_shared_queue", False)) else 0
        ctxs = []
        # every chain's prefill is issued on that chain's own stream: the chains' GEMMs / attention launches then fill each
        # other's last partial round of workgroups (a 10k-row x 1024-column GEMM is 632 tiles for 512 resident slots), and the
        # chain's decode graph follows on the same stream
        for i, (lo, hi) in enumerate(groups):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i] if self.PREFILL_ON_CHAIN_STREAMS else cur):
                ctxs.append(self._prepare(c_tokens[lo:hi], Lc[lo:hi], max_steps, sp_kw, slot=100 + i, row_offset=_row0 + lo, rows_total=rows_total,
                                          return_logits=return_logits, gate_lanes=lanes, use_graph=use_graph, shared_prefix=shared_prefix,
                                          force_tokens=None if force_tokens is None else torch.as_tensor(force_tokens)[lo:hi],
                                          z_tokens=None if z_tokens is None else torch.as_tensor(z_tokens)[lo:hi]))
        steps = min(c["steps"] for c in ctxs)
        if lanes:                      # re-arm the turnstile while nothing of the decode loop is in flight
            for s in streams:
                cur.wait_stream(s)
            self._sem.zero_()
            for c in ctxs:
                c["st"]["blk"].zero_()
            for s in streams:
                s.wait_stream(cur)
        if after_prefill is not None:
            for s in streams:
                cur.wait_stream(s)
            after_prefill()
            for s in streams:
                s.wait_stream(cur)
        done = 0
        while done < steps:
            n = min(check_every, steps - done) if stop_early else steps - done
            for _ in range(n):
                for c, s in zip(ctxs, streams):
                    with torch.cuda.stream(s):
                        if c["graph"] is not None:
                            c["graph"].replay()
                        else:
                            self.decode_step(c["st"], c["B"], c["sp"])
            done += n
            if stop_early:
                for s in streams:
                    cur.wait_stream(s)
                e_ = all(self._all_ended(c["st"], c["B"]) for c in ctxs)
                if (e_ if ended_reduce is None else bool(ended_reduce(e_))):
                    break
        for s in streams:
            cur.wait_stream(s)
        merged = {k: torch.cat([c["st"][k] for c in ctxs], 0) for k in ("seq", "len", "Lc", "logp")}
        res = dict(state=merged, steps=done)
        if return_logits:     # masked-logit history of every row, (B, max_steps, V) x 2 (parity tests of the chain interleave)
            res["logits_history"] = [torch.cat([c["hist"][i] for c in ctxs], 0) for i in range(2)]
        return res

    def _all_ended(self, st, B):
        seq, ln = st["seq"], st["len"].long()
        last = seq[torch.arange(B, device=self.dev), ln - 1].long()
        end = torch.tensor(self.end, device=self.dev)
        return bool((last == end).any(-1).all().item())   # shapeformer.py:112-115
