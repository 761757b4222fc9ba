"""Training step of the ShapeFormer transformer on MI355X.

Mirrors ShapeFormer.forward / shared_step / configure_optimizers (shapeformer/models/shapeformer/shapeformer.py:26-46,
132-207): teacher-forced logits over cz[:, :-1], mean of the two cross entropies on the outputs from index L_c-1 on
(end-token padding included), AdamW(lr, betas (0.9, 0.95)) with weight decay 0.01 on Linear weights only.
Every arithmetic step is a libsfmi (HIP) call: GEMM-shaped gradients go through the f32-MFMA GEMM on transposed
operands, the rest through csrc/train.hip.  Gradients live in ONE flat buffer so data-parallel training is a single
RCCL all-reduce (`torch.distributed`, backend "nccl" on ROCm) per step; the frozen VQDIF is replicated and excluded.

Dropout (mingpt.py:62-63,85,90,105,218,292: embd_pdrop on both stage inputs, attn_pdrop on the attention probabilities,
resid_pdrop after proj and after the MLP; 0.01 in shapenet_scale.yaml) is applied in `training_step` with counter-hash masks
(csrc/sfmi_common.h: element i of site s at step t is dropped iff hash_unit("dropout-<seed>-<t>-<s>")[i] < p), fused into the
producing kernels (attention forward / backward, GEMM epilogue); the CPU oracle builds the same masks, so train-mode
gradients are checked against its autograd exactly like the eval-mode ones.  torch's Philox stream cannot be matched.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _ru(x, m):
    return (x + m - 1) // m * m


def _concurrent_streams(dev, n, candidates=8):
    """n HIP streams that run concurrently WITH THE CURRENT STREAM and with each other.  The runtime multiplexes streams onto a few
    hardware queues; two streams on one queue run back to back.  Candidates are probed as gpt.CondTupleGPT._chain_streams probes them:
    one 200 us single-wavefront spin per stream, all at once - ~0.2 ms when every stream has a queue of its own, >= 0.4 ms otherwise.
    Falls back to unprobed streams when no set passes (correct either way: overlap is an optimisation)."""
    if n <= 0:
        return []
    cur = torch.cuda.current_stream(dev)
    ticks = 20000      # x 10 ns

    def overlap(ss):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        L.check(L.lib().sfmi_stream_spin(ticks, cur.cuda_stream), "sfmi_stream_spin")
        for st in ss:
            st.wait_event(e0)
            L.check(L.lib().sfmi_stream_spin(ticks, st.cuda_stream), "sfmi_stream_spin")
        for st in ss:
            cur.wait_stream(st)
        e1.record(cur)
        e1.synchronize()
        return e0.elapsed_time(e1) < 1.5 * ticks * 1e-5
    chosen, pool = [], [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    for st in pool:
        if len(chosen) == n:
            break
        if overlap(chosen + [st]) or overlap(chosen + [st]):
            chosen.append(st)
    return (chosen + [st for st in pool if st not in chosen])[:n]


class GPTTrainer:
    def __init__(self, gpt, lr=1e-5, betas=(0.9, 0.95), weight_decay=0.01, eps=1e-8, dist=None, pdrop=None, dropout_seed=0,
                 single_rank_collectives=False, grad_sync="ring", profile_waits=False, gemm="sk", overlap_param_gather=True, side_stream=True, fused_optimizer=True,
                 graph=False):
        """grad_sync: "ring" = per-bucket all-reduce, every rank updates every parameter; "rs_ag" = per-bucket reduce-scatter,
        AdamW on the rank's 1/N shard, all-gather of the updated parameters (dist.GradBuckets).  Same weights either way.
        gemm: "sk" = the work-balanced GEMM with fused GELU epilogues (csrc/sgemm_sk.hip, round 5); "tile" = one workgroup per
        128 x 128 tile + split-K reduce launches + separate GELU launches (csrc/sgemm.hip, rounds 2-4; kept as the cross-check).
        overlap_param_gather (rs_ag): the all-gather of bucket k's updated parameters is waited for right before the NEXT step's first
        kernel that reads them (embeddings, then block by block) instead of all at once after the optimizer.
        side_stream: the weight-gradient GEMMs and the per-block column reductions - needed by nobody before the optimizer - are issued
        on a second HIP stream, so that their workgroups fill the launch ramps and tails of the dependent chain (dX GEMMs, LayerNorm and
        attention backward) instead of queueing behind it; per-stream scratch, event-ordered, joined before the gradient collectives.
        graph: training_step() replays ONE captured hipGraph per (batch shape, train / eval mode) - forward, backward and the per-bucket AdamW
        on their three streams - instead of enqueueing ~860 launches from the host (the step is host-bound at the YAML's batch 1).  What
        changes between steps is read from device memory by the captured launches: the token tensors (static input buffers), every
        dropout site's seed and AdamW's bias corrections (one small host -> device copy per step, `_words`).  The first step of a shape
        runs eagerly (it also creates the lazily built scratch and tables), the second is captured.  Needs fused_optimizer and the "sk"
        GEMMs; with gradient collectives active (more than one rank) the step stays eager.  Same kernels in the same order: losses and
        weights are bit-identical to the eager step.
        fused_optimizer: training_step() updates a bucket (one transformer block / the heads / the embeddings) as soon as its gradients
        are final - gradient collective waited for per bucket, AdamW launched per bucket on the side stream - so the 9 GB of optimizer
        traffic (HBM-bound) runs under the backward pass of the blocks below (MFMA-bound) instead of after it.  Same update, same bits."""
        assert gemm in ("sk", "tile")
        # the constructor's own settings, so that a trainer can be rebuilt on new parameter tensors as it was (plugin.load_checkpoint)
        self._settings = dict(betas=tuple(betas), weight_decay=weight_decay, eps=eps, pdrop=pdrop, dropout_seed=dropout_seed,
                              single_rank_collectives=bool(single_rank_collectives), grad_sync=grad_sync, profile_waits=bool(profile_waits),
                              gemm=gemm, overlap_param_gather=bool(overlap_param_gather), side_stream=bool(side_stream),
                              fused_optimizer=bool(fused_optimizer), graph=bool(graph))
        self.use_graph = bool(graph)
        self._graphs, self._capture, self._graph_seen = {}, None, set()
        # per-step words read by the captured launches: [0:3] AdamW bias corrections and learning rate (f32 bits), [4:] dropout seeds of the sites in
        # capture order
        self._words = torch.zeros(4 + 256, device=gpt.dev, dtype=torch.int32)
        # a ring of pinned host mirrors: the host may run several replays ahead of the device, a mirror is rewritten only after its copy ran
        self._words_ring = [dict(buf=torch.zeros(4 + 256, dtype=torch.int32).pin_memory(), ev=None) for _ in range(4)] if graph else []
        self._words_next = 0
        self.gemm_algo, self.overlap_param_gather, self.fused_optimizer = gemm, bool(overlap_param_gather), bool(fused_optimizer)
        self._fused = False
        # the per-bucket optimizer of the fused step gets a stream of its own: it has to wait for the bucket's gradient collective, and
        # the weight-gradient GEMMs of the blocks below must not queue behind that wait
        extra = _concurrent_streams(gpt.dev, (1 if side_stream else 0) + (1 if (side_stream and fused_optimizer) else 0))
        self._side = extra[0] if side_stream else None
        self._opt_stream = extra[1] if (side_stream and fused_optimizer) else None
        self._on_side, self._keep, self._defer_side = False, [], None
        self.g, self.dev, self.D = gpt, gpt.dev, gpt.D
        # (embd_pdrop, resid_pdrop, attn_pdrop): the model's (CondTupleGPT ctor kwargs / YAML) unless given
        self.pdrop = tuple(float(v) for v in (pdrop if pdrop is not None else getattr(gpt, "pdrop", (0.0, 0.0, 0.0))))
        self.dropout_seed = dropout_seed
        self.lr, self.betas, self.wd, self.eps, self.dist = lr, betas, weight_decay, eps, dist
        self.step_count = 0
        g = gpt
        # ---- parameter table: (name, tensor, decay?) ; fused QKV is trained as one tensor (equivalent) -----------
        P = []
        for li, ly in enumerate(g.layers):
            p = f"L{li}."
            P += [(p + "ln1.w", ly.ln1[0], False), (p + "ln1.b", ly.ln1[1], False), (p + "wqkv", ly.wqkv, True),
                  (p + "bqkv", ly.bqkv, False), (p + "wproj", ly.wproj, True), (p + "bproj", ly.bproj, False),
                  (p + "ln2.w", ly.ln2[0], False), (p + "ln2.b", ly.ln2[1], False), (p + "wfc1", ly.wfc1, True),
                  (p + "bfc1", ly.bfc1, False), (p + "wfc2", ly.wfc2, True), (p + "bfc2", ly.bfc2, False)]
        for s in range(2):
            P += [(f"head{s}.ln.w", g.head_ln[s][0], False), (f"head{s}.ln.b", g.head_ln[s][1], False),
                  (f"head{s}.w", g.head_w[s], True)]
        P += [("E0", g.E[0], False), ("E1", g.E[1], False), ("Ex", g.Ex, False), ("pos_emb", g.pos_emb, False),
              ("cond_pos_emb", g.cond_pos_emb, False)]
        self.params = P
        n = sum(t.numel() for _, t, _ in P)
        self.flat_grad = torch.zeros(n, device=self.dev)
        self.grad, self.m, self.v = {}, {}, {}
        o = 0
        for name, t, _ in P:
            self.grad[name] = self.flat_grad[o:o + t.numel()].view(t.shape)
            o += t.numel()
        self.flat_m = torch.zeros(n, device=self.dev)
        self.flat_v = torch.zeros(n, device=self.dev)
        # gradient buckets for the overlapped all-reduce: one per block, one for both heads, one for the embeddings
        # (flat order = backward-completion order reversed: blocks 0..n-1, heads, embeddings)
        off, o = {}, 0
        for name, t, _ in P:
            off[name] = (o, o + t.numel())
            o += t.numel()
        rng = {}
        for li in range(len(g.layers)):
            rng[f"L{li}"] = (off[f"L{li}.ln1.w"][0], off[f"L{li}.bfc2"][1])
        rng["heads"] = (off["head0.ln.w"][0], off["head1.w"][1])
        rng["emb"] = (off["E0"][0], off["cond_pos_emb"][1])
        from .dist import GradBuckets
        self.buckets = GradBuckets(self.flat_grad, rng, dist, single_rank_collectives=single_rank_collectives, mode=grad_sync,
                                   profile_waits=profile_waits)
        self._sync = False
        self._emb_range = rng["emb"]
        import weakref
        gpt._param_sync = weakref.WeakMethod(self.finish_param_gather)      # every non-training reader of the parameters drains the in-flight gathers first (gpt._sync_params)
        # order in which a forward pass first reads the buckets' parameters (the order the rs_ag parameter gathers are launched in)
        self._fwd_order = ["emb"] + [f"L{li}" for li in range(len(g.layers))] + ["heads"]
        lib = L.lib()
        # scratch per stream that issues these launches ([0]: the main stream, [1]: the side stream)
        ns = 2 if self._side is not None else 1
        self._sk_slab = [torch.empty(lib.sfmi_sgemm_sk_slab_floats(), device=self.dev) for _ in range(ns)]      # stream-K partial tiles (134 MB)
        self._sk_cnt = [torch.zeros(1 << 20, device=self.dev, dtype=torch.int32) for _ in range(ns)]            # tickets: zeroed once, re-armed by the kernel
        self._cr_cnt = [torch.zeros(4096, device=self.dev, dtype=torch.int32) for _ in range(ns)]               # same for the column reductions
        self._cr_part = [None] * ns

    def settings(self):
        """The keyword arguments this trainer was built with (besides the model, `lr` and `dist`)."""
        return dict(self._settings)

    # ------------------------------------------------------------------ small wrappers
    def _f(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.float32)

    def _blas(self):
        """The plain GEMMs of the step (forward, dX = dY W, dW = dY^T X) run on csrc/sgemm.hip; SFMI_ROCBLAS=1 opts into the
        library sgemm (csrc/blas.hip, dlopen) where it can be bound."""
        if not hasattr(self, "_has_blas"):
            import os
            self._has_blas = os.environ.get("SFMI_ROCBLAS") == "1" and bool(L.lib().sfmi_blas_available())
        return self._has_blas

    def _sgemm(self, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, accumulate=False, bias=None, act=0, resid=None, drop=(0.0, 0), c2=None, aux=None):
        """One GEMM of the step.  "sk": csrc/sgemm_sk.hip (act 2 + c2: GELU with the pre-activation kept; act 3 + aux: times GELU'(aux)).
        "tile": csrc/sgemm.hip with split-K scratch for outputs of few tiles, the GELU forms as separate launches."""
        lib = L.lib()
        if self.gemm_algo == "sk":
            slab, cnt = self._sk_slab[int(self._on_side)], self._sk_cnt[int(self._on_side)]
            L.check(lib.sfmi_sgemm_sk_sd_f32(int(tA), int(tB), M, N, K, L.ptr(A), lda, L.ptr(B), ldb, L.ptr(C), L.ptr(c2), ldc, int(accumulate), L.ptr(bias), act,
                                             L.ptr(aux), L.ptr(resid), float(drop[0]), int(drop[1]), drop[2] if len(drop) > 2 else None, L.ptr(slab), slab.numel(),
                                             L.ptr(cnt), cnt.numel(), L.stream_ptr()), "sgemm_sk")
            return
        assert len(drop) < 3 or drop[2] is None, "the captured step needs the work-balanced GEMM (gemm='sk')"
        need = lib.sfmi_sgemm_mfma_splits(M, N, K) * M * N
        ws = None
        if need > M * N:
            key = "_sg_ws1" if self._on_side else "_sg_ws"
            if getattr(self, key, None) is None or getattr(self, key).numel() < need:
                setattr(self, key, torch.empty(need, device=self.dev))
            ws = getattr(self, key)
        if act == 2 and c2 is not None:          # pre-activation kept: GEMM -> c2, then GELU -> C
            assert resid is None and drop[0] == 0.0 and ldc == N
            L.check(lib.sfmi_sgemm_mfma_f32(int(tA), int(tB), M, N, K, L.ptr(A), lda, L.ptr(B), ldb, L.ptr(c2), ldc, int(accumulate), L.ptr(bias), 0, None,
                                            L.ptr(ws), ws.numel() if ws is not None else 0, 0.0, 0, L.stream_ptr()), "sgemm_mfma")
            L.check(lib.sfmi_gelu_f32(L.ptr(c2), L.ptr(C), M * N, L.stream_ptr()), "gelu")
            return
        if act == 3:                             # GEMM -> C, then C *= GELU'(aux)
            assert resid is None and bias is None and drop[0] == 0.0 and ldc == N
            L.check(lib.sfmi_sgemm_mfma_f32(int(tA), int(tB), M, N, K, L.ptr(A), lda, L.ptr(B), ldb, L.ptr(C), ldc, int(accumulate), None, 0, None,
                                            L.ptr(ws), ws.numel() if ws is not None else 0, 0.0, 0, L.stream_ptr()), "sgemm_mfma")
            L.check(lib.sfmi_gelu_bwd_f32(L.ptr(C), L.ptr(aux), L.ptr(C), M * N, L.stream_ptr()), "gelu_bwd")
            return
        L.check(lib.sfmi_sgemm_mfma_f32(int(tA), int(tB), M, N, K, L.ptr(A), lda, L.ptr(B), ldb, L.ptr(C), ldc, int(accumulate), L.ptr(bias),
                                        act, L.ptr(resid), L.ptr(ws), ws.numel() if ws is not None else 0, float(drop[0]), int(drop[1]), L.stream_ptr()), "sgemm_mfma")

    def _gemm(self, x, w, bias, resid, y, M, N, K, act=0, drop=(0.0, 0), c2=None):
        """y (M,N) = x (M,K) w^T (N,K) (+ bias) -> act -> dropout (+ resid)."""
        if c2 is None and drop[0] == 0.0 and self._blas() and M >= 256 and N % 4 == 0 and not (resid is not None and resid.data_ptr() == y.data_ptr() and act):
            L.check(L.lib().sfmi_gemm_blas_f32(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(resid), L.ptr(y), M, N, K, act, L.stream_ptr()), "gemm_blas")
            return
        if N % 4 == 0 and K % 4 == 0:
            self._sgemm(0, 1, M, N, K, x, K, w, K, y, N, bias=bias, act=act, resid=resid, drop=drop, c2=c2)
            return
        assert drop[0] == 0.0 and c2 is None
        L.check(L.lib().sfmi_gemm_f32(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(resid), L.ptr(y), M, N, K, act, 0, 0, L.stream_ptr()), "gemm")

    def _dx(self, dY, wname, w, M, N, K, gelu_aux=None):
        """dX (M,K) = dY (M,N) W (N,K)  [* GELU'(gelu_aux) elementwise: the backward of the activation that produced this layer's input]."""
        dx = self._f(M, K)
        if self._blas() and M >= 256 and gelu_aux is None:
            L.check(L.lib().sfmi_sgemm_f32(0, 0, M, K, N, 1.0, L.ptr(dY), N, L.ptr(w), K, 0.0, L.ptr(dx), K, L.stream_ptr()), "sgemm dx")
        else:      # W (N,K) is the (k,n)-stored right operand: no transposed copy
            self._sgemm(0, 0, M, K, N, dY, N, w, K, dx, K, act=3 if gelu_aux is not None else 0, aux=gelu_aux)
        return dx

    def _T(self, x, R, C, ld=None, Rpad=None):
        Rpad = Rpad or _ru(R, 16)
        out = self._f(C, Rpad)
        L.check(L.lib().sfmi_transpose_f32(L.ptr(x), L.ptr(out), R, C, ld or C, Rpad, L.stream_ptr()), "transpose")
        return out

    def _dW(self, dY, X, M, N, K, gname):
        """grad[gname] (N,K) (+)= dY^T (N,M) X (M,K)."""
        if self._blas() and M >= 256:      # dY^T X directly (transposed left operand), accumulating when asked to
            out = self.grad[gname]
            if N > K and not self._acc and M >= 1024:
                # tall outputs (4096 x 1024) hit a slow library kernel (86 vs 141 TFLOP/s for the wide 1024 x 4096 form):
                # form dW^T = X^T dY and transpose the 16 MB result
                tmp = self._f(K, N)
                L.check(L.lib().sfmi_sgemm_f32(1, 0, K, N, M, 1.0, L.ptr(X), K, L.ptr(dY), N, 0.0, L.ptr(tmp), N, L.stream_ptr()), "sgemm dW^T")
                L.check(L.lib().sfmi_transpose_f32(L.ptr(tmp), L.ptr(out), K, N, N, K, L.stream_ptr()), "transpose")
                return
            L.check(L.lib().sfmi_sgemm_f32(1, 0, N, K, M, 1.0, L.ptr(dY), N, L.ptr(X), K, 1.0 if self._acc else 0.0, L.ptr(out), K,
                                           L.stream_ptr()), "sgemm dW")
            return
        out = self.grad[gname]
        self._sgemm(1, 0, N, K, M, dY, N, X, K, out, K, accumulate=self._acc)   # dY^T X with dY / X read in place (no transposes)

    def _aside(self, fn, *tensors):
        """Run `fn()` (launches that only the optimizer waits for: weight-gradient GEMMs, column reductions) on the side stream, ordered
        after everything the main stream has enqueued so far.  `tensors`: what those launches read - kept alive until the streams are
        joined (their memory must not be handed to a later main-stream allocation while the side stream still reads it)."""
        if self._side is None:
            return fn()
        if self._defer_side is not None:      # captured step: a block's side work is issued as ONE fork at the block's end (_ready)
            self._defer_side.append(fn)
            self._keep.extend(tensors)
            return None
        ev = torch.cuda.Event()
        ev.record()
        self._side.wait_event(ev)
        self._keep.extend(tensors)
        self._on_side = True
        try:
            with torch.cuda.stream(self._side):
                return fn()
        finally:
            self._on_side = False

    def _join_side(self):
        """The main stream waits for the side streams (before gradients / updated parameters are consumed) and the kept tensors go."""
        if self._side is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self._side)
            if self._opt_stream is not None and self._capture is None:      # (the captured step never forks onto the optimizer stream)
                cur.wait_stream(self._opt_stream)
            self._keep.clear()

    def _ln_rows(self, dy, x, gamma, dres, M, drop=(0.0, 0)):
        """LayerNorm backward, row part: -> (dx = dLN/dx (+ dres), stats (M,2), dropped); the parameter sums join the block's column
        reduction.  drop = (p, seed) with p > 0: `dropped` = nn.Dropout(dx) under that site's mask, from the same launch (the gradient
        the next GEMMs need when the forward dropped this tensor); else `dropped` is dx itself."""
        dx, stats = self._f(M, self.D), self._f(M, 2)
        dx2 = self._f(M, self.D) if drop[0] > 0.0 else None
        L.check(L.lib().sfmi_layernorm_bwd_rows_drop_sd_f32(L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(dres), L.ptr(dx), L.ptr(stats), L.ptr(dx2),
                                                            float(drop[0]), int(drop[1]), drop[2] if len(drop) > 2 else None, M, self.D,
                                                            L.stream_ptr()), "ln_bwd_rows")
        return dx, stats, (dx2 if dx2 is not None else dx)

    def _col_reduce(self, jobs, M):
        """All bias / LayerNorm-parameter gradients of a block in ONE launch (csrc/train.hip:col_reduce_kernel).  jobs: ("b", dY (M,N), N,
        grad name) | ("ln", dy, x, stats, dgamma name, dbeta name).  Writes (accumulates when the step accumulates)."""
        import ctypes as C
        lib = L.lib()
        n = len(jobs)
        kind, a, x, st, o, o2, N, ld = [], [], [], [], [], [], [], []
        for j in jobs:
            if j[0] == "b":
                _, dY, cols, gname = j
                kind.append(0); a.append(dY.data_ptr()); x.append(0); st.append(0); o.append(self.grad[gname].data_ptr()); o2.append(0)
                N.append(cols); ld.append(cols)
            else:
                _, dy, xin, stats, gw, gb = j
                kind.append(1); a.append(dy.data_ptr()); x.append(xin.data_ptr()); st.append(stats.data_ptr())
                o.append(self.grad[gw].data_ptr()); o2.append(self.grad[gb].data_ptr()); N.append(self.D); ld.append(self.D)
        need = lib.sfmi_col_reduce_part_floats(M, sum(N))
        si = int(self._on_side)
        if lib.sfmi_col_reduce_slices(M) > 1 and (self._cr_part[si] is None or self._cr_part[si].numel() < need):
            self._cr_part[si] = torch.empty(need, device=self.dev)
        part, cnt = self._cr_part[si], self._cr_cnt[si]
        IA, PA = C.c_int * n, C.c_void_p * n
        # the launcher copies the tables into the kernel argument before it returns: the ctypes arrays may die afterwards
        L.check(lib.sfmi_col_reduce_f32(n, IA(*kind), PA(*a), PA(*x), PA(*st), PA(*o), PA(*o2), IA(*N), IA(*ld), M, int(self._acc),
                                        L.ptr(part), part.numel() if part is not None else 0, L.ptr(cnt), cnt.numel(), L.stream_ptr()), "col_reduce")

    def _scatter(self, dx, idx, table, M, accumulate=True):
        rows = self.grad[table].shape[0]
        acc = torch.zeros(rows * self.D, device=self.dev, dtype=torch.int64)
        L.check(L.lib().sfmi_embed_scatter_f32(L.ptr(dx), L.ptr(idx), L.ptr(acc), M, self.D, L.stream_ptr()), "scatter")
        L.check(L.lib().sfmi_fixed_to_float_f32(L.ptr(acc), L.ptr(self.grad[table]), rows * self.D, int(accumulate), L.stream_ptr()), "fix2f")

    # ------------------------------------------------------------------ forward + backward
    @torch.no_grad()
    def _ready(self, name):
        """Bucket `name` is final once the launches enqueued so far - on BOTH streams - have run: its collective is launched from the
        side stream after that stream has been ordered behind the main one (RCCL orders a collective after the launching stream)."""
        def go():
            if self._sync:
                self.buckets.ready(name)
            if self._fused:
                if self._opt_stream is None or self._capture is not None:      # captured step (no collective to wait for): AdamW follows on the side stream
                    self._opt_bucket(name)
                else:      # optimizer stream: after the side stream's work so far (this bucket's weight gradients, its collective launch)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._opt_stream.wait_event(ev)
                    with torch.cuda.stream(self._opt_stream):
                        self._opt_bucket(name)
        if self._defer_side is not None:
            # one cross-stream dependency per bucket instead of five: a fork of a captured hipGraph costs ~20 us of latency on its branch
            # (profiles/r04_b16_experiments.md), and with five per block the replayed step was SLOWER than the eager one (16.8 vs 14.7 ms)
            queued, self._defer_side = self._defer_side, None
            try:
                self._aside(lambda: [f() for f in queued] + [go()])
            finally:
                self._defer_side = []
            return
        if self._side is not None and ((self._sync and self.buckets.active) or self._fused):
            self._aside(go)
        else:
            go()

    @torch.no_grad()
    def _opt_bucket(self, name):
        """The fused step's update of ONE bucket (called where the bucket's gradients are final, on the side stream when there is one):
        wait for its gradient collective, AdamW over the bucket (rs_ag: over this rank's slice, the updated values also go into the
        flat buffer), rs_ag: launch the all-gather of the bucket's parameters.  step_count was advanced by training_step()."""
        bk = self.buckets
        bk.wait_bucket(name)
        sharded = bk.active and bk.mode == "rs_ag"
        key = (name, sharded)
        if not hasattr(self, "_opt_tabs"):
            self._opt_tabs = {}
        if key not in self._opt_tabs:
            self._opt_tabs[key] = self._chunk_table([bk.shard(name) if sharded else bk.ranges[name]])
        tb = self._opt_tabs[key]
        L.check(L.lib().sfmi_adamw_multi_shard_bc_f32(L.ptr(tb["p"]), L.ptr(tb["foff"]), L.ptr(tb["wd"]), L.ptr(tb["ct"]), L.ptr(tb["co"]), L.ptr(tb["cl"]),
                                                      tb["n"], L.ptr(self.flat_grad), L.ptr(self.flat_m), L.ptr(self.flat_v), self.lr, self.betas[0],
                                                      self.betas[1], self.eps, self.step_count,
                                                      self._words.data_ptr() if self._capture is not None else None,      # captured step: bias corrections from the step words
                                                      L.ptr(self.flat_grad) if sharded else None, L.stream_ptr()), "adamw_multi")
        if sharded and bk.sharded(name):
            if self.overlap_param_gather:
                bk.launch_param_gather(name)
            else:
                w = bk._all_gather_bucket(self.flat_grad, name)
                w.wait()
                fb = self._bucket_tab(name)
                L.check(L.lib().sfmi_unflatten_multi_f32(L.ptr(fb["p"]), L.ptr(fb["foff"]), L.ptr(fb["ct"]), L.ptr(fb["co"]), L.ptr(fb["cl"]), fb["n"],
                                                         L.ptr(self.flat_grad), L.stream_ptr()), "unflatten_multi")

    @torch.no_grad()
    def _param_ready(self, name):
        """rs_ag with overlapped parameter gathers: make the compute stream wait for the all-gather of bucket `name` (launched by the
        previous optimizer_step) and copy its gathered parameters out of the flat buffer into the tensors - right before the first
        kernel that reads them.  No-op when nothing is pending for the bucket."""
        if self.buckets.wait_params(name):
            tb = self._bucket_tab(name)
            L.check(L.lib().sfmi_unflatten_multi_f32(L.ptr(tb["p"]), L.ptr(tb["foff"]), L.ptr(tb["ct"]), L.ptr(tb["co"]), L.ptr(tb["cl"]), tb["n"],
                                                     L.ptr(self.flat_grad), L.stream_ptr()), "unflatten_multi")

    def _bucket_tab(self, name):
        if not hasattr(self, "_bucket_tabs"):
            self._bucket_tabs = {}
        if name not in self._bucket_tabs:
            self._bucket_tabs[name] = self._chunk_table([self.buckets.ranges[name]])
        return self._bucket_tabs[name]

    @torch.no_grad()
    def finish_param_gather(self):
        """Drain every pending parameter all-gather (rs_ag): call before anything but the next training forward reads the parameter
        tensors (state_dict / checkpoint, sampling, evaluation).  Every rank reaches it at the same point of its program."""
        for name in self.buckets.params_in_flight():
            self._param_ready(name)

    def loss_and_grad(self, c_indices, z_indices, accumulate=False, sync=False, dropout_key=None, _fused=False):
        """-> loss (0-dim tensor).  Gradients of every parameter are left in self.grad[name] (flat buffer).
        dropout_key=None: eval-mode graph (no dropout); a string: train mode, the mask of site s is hash "dropout-<key>-<s>".
        sync=True (last micro-step of a data-parallel step): each block's gradient bucket is all-reduced as soon as it
        is final, under the backward kernels of the remaining blocks (dist.GradBuckets); finish with all_reduce_grads()."""
        g, D, dev, lib = self.g, self.D, self.dev, L.lib()
        self._acc, self._sync, self._fused = accumulate, sync, bool(_fused)
        self._param_ready("emb")       # rs_ag: the embeddings' gathered parameters leave the flat buffer before it is written again
        self._param_ready("L0")        # the embedding kernel also applies block 0's first LayerNorm
        if not accumulate:
            # every gradient below is WRITTEN (GEMM / column-reduction epilogues), except the embedding tables: E0 is accumulated
            # twice and the positional tables only receive the rows the batch touches - those 57 MB are zeroed, not the 1.3 GB buffer
            lo, hi = self._emb_range
            if getattr(self, "debug_poison_grads", False):
                # debug: the claim above is checked - everything outside the embedding range starts as NaN and must have been overwritten
                # (not added to) by the end of the backward pass; with grad_sync "rs_ag" the buffer holds PARAMETER values at this point,
                # which a conditionally written gradient would silently absorb
                self.flat_grad[:lo].fill_(float("nan"))
                self.flat_grad[hi:].fill_(float("nan"))
            self.flat_grad[lo:hi].zero_()
        c = torch.as_tensor(c_indices).to(dev, torch.int32)
        z = torch.as_tensor(z_indices).to(dev, torch.int32)
        cz = torch.cat([c, z], 1).contiguous()
        B, Lc, Lz = c.shape[0], c.shape[1], z.shape[1]
        Lq = Lc + Lz - 1
        M = B * Lq
        assert Lq <= g.Lmax
        st = dict(seq=torch.zeros(B, g.Lmax + 1, 2, device=dev, dtype=torch.int32),
                  Lc=torch.full((B,), Lc, device=dev, dtype=torch.int32), len=torch.full((B,), Lq, device=dev, dtype=torch.int32),
                  nval=torch.full((B,), Lq, device=dev, dtype=torch.int32), extra=None,
                  extra_out=torch.zeros(M, device=dev, dtype=torch.int32))
        st["seq"][:, :Lq + 1] = cz
        kv = self._f(2, B, g.Lmax + 1, D)   # prefill attention also fills a KV cache; scratch here
        from .weights import _fnv1a32
        p_embd, p_resid, p_attn = self.pdrop if dropout_key is not None else (0.0, 0.0, 0.0)

        def site(p, name):          # (p, seed, device address of the seed or None) of one dropout site
            if p <= 0.0:
                return (0.0, 0, None)
            if self._capture is not None:      # captured step: the launch reads this site's seed from the step words at run time
                idx = self._capture["sites"].setdefault(name, len(self._capture["sites"]))
                assert idx < self._words.numel() - 4, "too many dropout sites for the step-word table"
                return (p, 0, self._words.data_ptr() + 4 * (4 + idx))
            return (p, _fnv1a32(f"dropout-{dropout_key}-{name}"), None)

        def drop_(x, ps, out=None):  # elementwise nn.Dropout (forward == backward): out = x * mask / (1 - p)
            if ps[0] == 0.0:
                return x
            out = out if out is not None else self._f(*x.shape)
            L.check(lib.sfmi_dropout_sd_f32(L.ptr(x), L.ptr(out), x.numel(), ps[0], ps[1], ps[2], L.stream_ptr()), "dropout")
            return out
        # ---- forward, keeping what the backward needs --------------------------------------------------------
        saved = []
        resid = self._f(M, D)
        xn = self._f(M, D)
        if p_embd > 0.0:     # x = drops[0](embeddings) (mingpt.py:292)
            g._embed(st, B, Lq, resid, None, None)
            drop_(resid, site(p_embd, "emb0"), out=resid)
            g._rowprep(resid, None, None, 0, M, None, xn, g.layers[0].ln1)
        else:
            g._embed(st, B, Lq, resid, xn, g.layers[0].ln1)
        head_in = {}
        for li, ly in enumerate(g.layers):
            s = dict(x_in=resid, xn1=xn)
            qkv = self._f(M, 3 * D)
            self._gemm(xn, ly.wqkv, ly.bqkv, None, qkv, M, 3 * D, D)
            y = self._f(M, D)
            lse_l = self._f(B, g.H, Lq)        # row log-sum-exps of the scaled scores: the attention backward starts from them
            if D == 64 * g.H and B * g.H * ((Lq + 63) // 64) <= 128:
                # too few 64-row tiles for 256 CUs (the YAML's batch 1): 32-row tiles x two key-block groups per workgroup
                L.check(lib.sfmi_attn_train_fwd_small_sd_f32(L.ptr(qkv), L.ptr(y), L.ptr(lse_l), B, Lq, D, g.H, *site(p_attn, f"L{li}.attn"),
                                                             L.stream_ptr()), "attn")
            else:
                L.check(lib.sfmi_gpt_attn_prefill_lse_sd_f32(L.ptr(qkv), L.ptr(kv[0]), L.ptr(kv[1]), L.ptr(st["nval"]), L.ptr(y), B, Lq, D,
                                                             g.H, g.Lmax + 1, None, *site(p_attn, f"L{li}.attn"), L.ptr(lse_l), L.stream_ptr()), "attn")
            r1 = self._f(M, D)
            self._gemm(y, ly.wproj, ly.bproj, resid, r1, M, D, D, drop=site(p_resid, f"L{li}.proj"))
            xn2 = self._f(M, D)
            g._rowprep(r1, None, None, 0, M, None, xn2, ly.ln2)
            hpre, h = self._f(M, 4 * D), self._f(M, 4 * D)
            self._gemm(xn2, ly.wfc1, ly.bfc1, None, h, M, 4 * D, D, act=2, c2=hpre)     # h = GELU(hpre), both kept (mingpt.py:102-103)
            r2 = self._f(M, D)
            self._gemm(h, ly.wfc2, ly.bfc2, r1, r2, M, D, 4 * D, drop=site(p_resid, f"L{li}.mlp"))
            s.update(qkv=qkv, y=y, r1=r1, xn2=xn2, hpre=hpre, h=h, lse=lse_l)
            saved.append(s)
            resid = r2
            last = li + 1 == len(g.layers) or g.layers[li + 1].stage != ly.stage
            if last:
                head_in[ly.stage] = resid
            if li + 1 < len(g.layers):
                nxt = g.layers[li + 1]
                self._param_ready(f"L{li + 1}")      # the next block's parameters (its ln1 is applied right here)
                xn = self._f(M, D)
                if nxt.stage != ly.stage:
                    r3 = self._f(M, D)
                    if p_embd > 0.0:     # x = drops[1](h20 + tok_embs[0](target pos)) (mingpt.py:292,295)
                        g._rowprep(resid, None, None, 0, M, r3, None, None, Eadd=g.E[0], P=Lq, st=st)
                        drop_(r3, site(p_embd, "emb1"), out=r3)
                        g._rowprep(r3, None, None, 0, M, None, xn, nxt.ln1)
                    else:
                        g._rowprep(resid, None, None, 0, M, r3, xn, nxt.ln1, Eadd=g.E[0], P=Lq, st=st)
                    resid = r3
                else:
                    g._rowprep(resid, None, None, 0, M, None, xn, nxt.ln1)
        # ---- heads, loss, dlogits ---------------------------------------------------------------------------------
        tgt = cz[:, 1:, :].contiguous()     # targets of every input position (only t >= Lc-1 are active)
        loss = torch.zeros((), device=dev)
        d_head, head_jobs = {}, []
        scale = 1.0 / (2.0 * B * Lz)
        self._param_ready("heads")
        for s in range(2):
            xnh = self._f(M, D)
            g._rowprep(head_in[s], None, None, 0, M, None, xnh, g.head_ln[s])
            lg = self._f(M, g.Vpad)
            self._gemm(xnh, g.head_w_pad[s], None, None, lg, M, g.Vpad, D)
            rows, dlg = self._f(M), self._f(M, g.Vpad)
            t_s = tgt[..., s].contiguous().view(-1)
            L.check(lib.sfmi_ce_fwd_bwd_f32(L.ptr(lg), L.ptr(t_s), L.ptr(rows), L.ptr(dlg), M, g.V, g.Vpad, Lq, Lc - 1, scale,
                                            L.stream_ptr()), "ce")
            loss = loss + rows.sum() * scale     # scalar for logging only
            # head backward: dW_h = dlg^T xnh ; dxnh = dlg W_h ; LN backward
            Mp = _ru(M, 16)
            dlgT, xnhT = self._T(dlg, M, g.Vpad, Rpad=Mp), self._T(xnh, M, D, Rpad=Mp)
            gh = self.grad[f"head{s}.w"]                                  # rows >= V of dlg^T are zero: write (V,D) directly
            self._gemm(dlgT, xnhT, None, gh if accumulate else None, gh, g.V, D, Mp)
            dxnh = self._f(M, D)
            if self.gemm_algo == "sk":      # dX = dlg W_h with W_h (Vpad, D) read in place as the (k, n)-stored operand: no 16 MB transpose per head
                self._sgemm(0, 0, M, D, g.Vpad, dlg, g.Vpad, g.head_w_pad[s], D, dxnh, D)
            else:
                whT = self._T(g.head_w_pad[s], g.Vpad, D, Rpad=g.Vpad)      # (D, Vpad)
                self._gemm(dlg, whT, None, None, dxnh, M, D, g.Vpad)
            d_head[s], hst, _ = self._ln_rows(dxnh, head_in[s], g.head_ln[s][0], None, M)
            head_jobs.append(("ln", dxnh, head_in[s], hst, f"head{s}.ln.w", f"head{s}.ln.b"))
        self._col_reduce(head_jobs, M)
        self._ready("heads")
        # ---- backward through the blocks -----------------------------------------------------------------------------
        dr = d_head[1]
        dm_pre = None                     # nn.Dropout(dr) for the current block's MLP site, when the block above already produced it
        delta = self._f(B, g.H, Lq)       # row sums of dO * O (scratch of sfmi_attn_bwd_lse_f32)
        for li in range(len(g.layers) - 1, -1, -1):
            ly, s, p = g.layers[li], saved[li], f"L{li}."
            if li + 1 < len(g.layers) and g.layers[li + 1].stage != ly.stage:
                # stage boundary (mingpt.py:294): x1 = h20 + E0[target pos]  ->  dE0 += scatter(dr) ; dh20 = dr + head-0 path
                dr = drop_(dr, site(p_embd, "emb1"))          # through drops[1]
                self._scatter(dr, tgt[..., 0].contiguous().view(-1), "E0", M, accumulate=True)
                dsum = self._f(M, D)
                L.check(lib.sfmi_add_f32(L.ptr(dr), L.ptr(d_head[0]), L.ptr(dsum), M * D, L.stream_ptr()), "add")
                dr, dm_pre = dsum, None
            # fc2 (the GELU backward rides in the epilogue of dX)
            # gradient of the MLP output before its dropout (residual path: dr); the block above emitted it with its LayerNorm backward
            dm = dm_pre if dm_pre is not None else drop_(dr, site(p_resid, f"L{li}.mlp"))
            self._aside(lambda: self._dW(dm, s["h"], M, D, 4 * D, p + "wfc2"), dm, s["h"])
            dhpre = self._dx(dm, p + "wfc2", ly.wfc2, M, D, 4 * D, gelu_aux=s["hpre"])
            # fc1
            self._aside(lambda: self._dW(dhpre, s["xn2"], M, 4 * D, D, p + "wfc1"), dhpre, s["xn2"])
            dxn2 = self._dx(dhpre, p + "wfc1", ly.wfc1, M, 4 * D, D)
            dr1, st2, dp = self._ln_rows(dxn2, s["r1"], ly.ln2[0], dr, M, drop=site(p_resid, f"L{li}.proj"))
            # proj (dp = the gradient through the projection's dropout, from the same launch)
            self._aside(lambda: self._dW(dp, s["y"], M, D, D, p + "wproj"), dp, s["y"])
            dy = self._dx(dp, p + "wproj", ly.wproj, M, D, D)
            # attention
            dqkv = self._f(M, 3 * D)
            L.check(lib.sfmi_attn_bwd_lse_sd_f32(L.ptr(s["qkv"]), L.ptr(s["y"]), L.ptr(dy), L.ptr(s["lse"]), L.ptr(delta), L.ptr(dqkv), B, Lq, D, g.H,
                                                 *site(p_attn, f"L{li}.attn"), L.stream_ptr()), "attn_bwd")
            # qkv
            self._aside(lambda: self._dW(dqkv, s["xn1"], M, 3 * D, D, p + "wqkv"), dqkv, s["xn1"])
            dxn1 = self._dx(dqkv, p + "wqkv", ly.wqkv, M, 3 * D, D)
            # ... and through the MLP dropout of the block below, unless a stage boundary rewrites dr first
            below = li > 0 and g.layers[li - 1].stage == ly.stage
            dr, st1, dm_pre = self._ln_rows(dxn1, s["x_in"], ly.ln1[0], dr1, M, drop=site(p_resid, f"L{li - 1}.mlp") if below else (0.0, 0, None))
            if not below:
                dm_pre = None
            # the block's four bias gradients and two LayerNorm parameter gradients: one launch
            jobs = [("b", dm, D, p + "bfc2"), ("b", dhpre, 4 * D, p + "bfc1"), ("b", dp, D, p + "bproj"), ("b", dqkv, 3 * D, p + "bqkv"),
                    ("ln", dxn2, s["r1"], st2, p + "ln2.w", p + "ln2.b"), ("ln", dxn1, s["x_in"], st1, p + "ln1.w", p + "ln1.b")]
            self._aside(lambda: self._col_reduce(jobs, M), dxn2, dxn1, s["r1"], s["x_in"], st1, st2)
            saved[li] = None
            self._ready(f"L{li}")     # this block's 50 MB of gradients are final: all-reduce under the next blocks' backward
        # ---- embeddings (mingpt.py:256-286): E0[pos] + E1[val] + Ex[extra] + positional --------------------------------
        idx = cz[:, :Lq, :]
        dr = drop_(dr, site(p_embd, "emb0"))                  # through drops[0]
        self._scatter(dr, idx[..., 0].contiguous().view(-1), "E0", M)
        self._scatter(dr, idx[..., 1].contiguous().view(-1), "E1", M, accumulate=accumulate)
        self._scatter(dr, st["extra_out"], "Ex", M, accumulate=accumulate)   # AR_N index as used by the forward
        # positional tables: row t of every batch item -> cond_pos_emb[t] (t < Lc) / pos_emb[t - Lc]  (sum over batch)
        gc, gp = self.grad["cond_pos_emb"], self.grad["pos_emb"]
        L.check(lib.sfmi_colsum_f32(L.ptr(dr), L.ptr(gc), B, Lc * D, Lq * D, int(accumulate), L.stream_ptr()), "colsum")
        if Lq > Lc:
            L.check(lib.sfmi_colsum_f32(dr.data_ptr() + Lc * D * 4, L.ptr(gp), B, (Lq - Lc) * D, Lq * D, int(accumulate),
                                        L.stream_ptr()), "colsum")
        self._ready("emb")
        self._join_side()      # the gradients of every block are complete on the main stream from here on
        if getattr(self, "debug_poison_grads", False) and not accumulate and not self._fused and not (self._sync and self.buckets.active):
            assert bool(torch.isfinite(self.flat_grad).all()), "a gradient range was not fully written by the backward pass"
        return loss

    # ------------------------------------------------------------------ optimizer / data parallel
    @torch.no_grad()
    def all_reduce_grads(self):
        """DDP gradient synchronisation (mean over ranks): waits for the per-block collectives that `loss_and_grad(...,
        sync=True)` launched under the backward pass, and launches any bucket that was not (e.g. sync=False)."""
        return self.buckets.finish()

    @torch.no_grad()
    def optimizer_step(self):
        """torch.optim.AdamW over the two reference groups (shapeformer.py:198-206)."""
        self.step_count += 1
        if not hasattr(self, "_adam_tab"):      # device tables for the one-launch update (built once)
            self._adam_tab = self._chunk_table([(0, self.flat_grad.numel())])
        tb = self._adam_tab
        sharded = self.buckets.active and self.buckets.mode == "rs_ag"
        if sharded:
            # optimizer sharding: after the reduce-scatter this rank holds the mean gradient of ITS slice of every bucket only;
            # it updates those elements (and their moments), writes the new parameter values over the consumed gradients in the
            # flat buffer, all-gathers the buckets in place and copies the gathered parameters back into the tensors
            if not hasattr(self, "_adam_tab_shard"):
                self._adam_tab_shard = self._chunk_table(self.buckets.shard_ranges())
            tb = self._adam_tab_shard
        L.check(L.lib().sfmi_adamw_multi_shard_f32(L.ptr(tb["p"]), L.ptr(tb["foff"]), L.ptr(tb["wd"]), L.ptr(tb["ct"]), L.ptr(tb["co"]), L.ptr(tb["cl"]),
                                                   tb["n"], L.ptr(self.flat_grad), L.ptr(self.flat_m), L.ptr(self.flat_v), self.lr, self.betas[0],
                                                   self.betas[1], self.eps, self.step_count, L.ptr(self.flat_grad) if sharded else None,
                                                   L.stream_ptr()), "adamw_multi")
        if sharded:
            # NOTE: from here on the flat buffer (and the self.grad[name] views into it) holds PARAMETER values, not gradients
            if self.overlap_param_gather:
                # one all-gather per bucket, launched in the order the next forward reads the parameters; each is waited for (and
                # unflattened) right before its first use: _param_ready() in loss_and_grad / finish_param_gather()
                self.buckets.launch_param_gathers(self._fwd_order)
            else:
                self.buckets.all_gather_params()
                fb = self._adam_tab
                L.check(L.lib().sfmi_unflatten_multi_f32(L.ptr(fb["p"]), L.ptr(fb["foff"]), L.ptr(fb["ct"]), L.ptr(fb["co"]), L.ptr(fb["cl"]), fb["n"],
                                                         L.ptr(self.flat_grad), L.stream_ptr()), "unflatten_multi")
        self.g.mark_decode_weights_stale()  # LN-folded / fragment-packed decode weights are rebuilt at the next decode use

    def _chunk_table(self, ranges, CH=16384):
        """Device chunk tables of sfmi_adamw_multi_shard_f32 / sfmi_unflatten_multi_f32 for the elements of the flat ranges
        `ranges` (sorted, disjoint): every parameter tensor's overlap with a range, cut into chunks of <= CH elements."""
        ptrs, foff, wd, ct, co, cl = [], [], [], [], [], []
        o = 0
        for ti, (name, t, decay) in enumerate(self.params):
            n = t.numel()
            ptrs.append(t.data_ptr()); foff.append(o); wd.append(self.wd if decay else 0.0)
            for lo, hi in ranges:
                a, b = max(lo, o), min(hi, o + n)
                for c0 in range(a, b, CH):
                    ct.append(ti); co.append(c0 - o); cl.append(min(CH, b - c0))
            o += n
        dev = self.dev
        return dict(p=torch.tensor(ptrs, dtype=torch.int64, device=dev), foff=torch.tensor(foff, dtype=torch.int64, device=dev),
                    wd=torch.tensor(wd, dtype=torch.float32, device=dev), ct=torch.tensor(ct, dtype=torch.int32, device=dev),
                    co=torch.tensor(co, dtype=torch.int64, device=dev), cl=torch.tensor(cl, dtype=torch.int32, device=dev), n=len(ct))

    def _sharded(self):
        return self.buckets.active and self.buckets.mode == "rs_ag"

    def optimizer_state(self):
        """Resume state, LOCAL (no collective): AdamW moments (flat, in parameter-table order) and the step count.  With the
        sharded optimizer (grad_sync "rs_ag") a rank only holds the moments of its own slices: the dict then carries `shard_ranges`
        (+ rank / world) and the other elements are zero; `gather_optimizer_state()` is the collective that completes it."""
        st = dict(step=self.step_count, exp_avg=self.flat_m.detach().cpu(), exp_avg_sq=self.flat_v.detach().cpu(),
                  names=[n for n, _, _ in self.params])
        if self._sharded():
            st.update(shard_ranges=[tuple(r) for r in self.buckets.shard_ranges()], rank=self.buckets.rank, world=self.buckets.world)
        return st

    def gather_optimizer_state(self):
        """The complete optimizer state on EVERY rank (grad_sync "rs_ag": an all-gather of both moment buffers - a collective, to be
        called by all ranks at the same point of the training loop, e.g. once per checkpoint interval; the result can be resumed in
        either synchronisation mode, on any world size).  Without sharding this is optimizer_state()."""
        if not self._sharded():
            return self.optimizer_state()
        m, v = self.flat_m.clone(), self.flat_v.clone()
        self.buckets.all_gather_flat(m)
        self.buckets.all_gather_flat(v)
        return dict(step=self.step_count, exp_avg=m.cpu(), exp_avg_sq=v.cpu(), names=[n for n, _, _ in self.params])

    def load_optimizer_state(self, st):
        assert st["names"] == [n for n, _, _ in self.params], "optimizer state does not match this parameter table"
        if st.get("shard_ranges") is not None:       # a rank-local state of a sharded run: only valid for the same rank of the same sharding
            if not self._sharded() or (st["rank"], st["world"]) != (self.buckets.rank, self.buckets.world) or \
                    [tuple(r) for r in st["shard_ranges"]] != [tuple(r) for r in self.buckets.shard_ranges()]:
                raise ValueError("load_optimizer_state: a rank-local shard of a sharded optimizer can only be resumed by the same rank of the same "
                                 "sharding; save gather_optimizer_state() to resume elsewhere")
        self.step_count = int(st["step"])
        self.flat_m.copy_(st["exp_avg"].to(self.dev))
        self.flat_v.copy_(st["exp_avg_sq"].to(self.dev))

    @torch.no_grad()
    def training_step(self, c_indices, z_indices):
        """One optimizer step in TRAIN mode (dropout on when the model's pdrop > 0; every rank / step draws its own masks).
        fused_optimizer: the gradient collectives are consumed and the parameters updated bucket by bucket INSIDE the backward pass
        (see the constructor); else loss_and_grad -> all_reduce_grads -> optimizer_step.  The result is the same bit for bit."""
        rank = self.dist.get_rank() if (self.dist is not None and self.dist.is_initialized()) else 0
        key = f"{self.dropout_seed}-{rank}-{self.step_count}" if any(self.pdrop) else None
        if self.use_graph and self.fused_optimizer and self.gemm_algo == "sk" and not self.buckets.active and not self._blas():
            return self._training_step_graphed(c_indices, z_indices, key)
        if not self.fused_optimizer:
            loss = self.loss_and_grad(c_indices, z_indices, sync=True, dropout_key=key)
            self.all_reduce_grads()
            self.optimizer_step()
            return loss
        self.step_count += 1
        try:
            loss = self.loss_and_grad(c_indices, z_indices, sync=True, dropout_key=key, _fused=True)
        except BaseException:
            self.step_count -= 1       # no update was completed under this count
            raise
        finally:
            self._fused = False
        self.buckets.end_step()
        self.g.mark_decode_weights_stale()
        return loss

    # ------------------------------------------------------------------ the step as one hipGraph
    @torch.no_grad()
    def _training_step_graphed(self, c_indices, z_indices, key):
        """training_step() through a captured hipGraph (see the constructor).  `key`: this step's dropout key or None (eval-mode graph)."""
        from .weights import _fnv1a32
        dev = self.dev
        c = torch.as_tensor(c_indices)
        z = torch.as_tensor(z_indices)
        gkey = (tuple(c.shape), tuple(z.shape), key is not None, int(L.lib().sfmi_tune_generation()))      # (the learning rate is a step word, not a captured argument)
        G = self._graphs.get(gkey)
        if G is None and gkey not in self._graph_seen:
            # first step of this shape: eager.  It builds what the step creates lazily OUTSIDE a capture pool (per-bucket AdamW tables,
            # column-reduction scratch, kernel attributes) and is a full, ordinary step
            self._graph_seen.add(gkey)
            self.step_count += 1
            try:
                loss = self.loss_and_grad(c, z, sync=True, dropout_key=key, _fused=True)
            except BaseException:
                self.step_count -= 1
                raise
            finally:
                self._fused = False
            self.buckets.end_step()
            self.g.mark_decode_weights_stale()
            return loss
        self.step_count += 1
        if G is None:
            cb = torch.zeros(tuple(c.shape), device=dev, dtype=torch.int32)
            zb = torch.zeros(tuple(z.shape), device=dev, dtype=torch.int32)
            self._capture = dict(sites={})
            self._defer_side = [] if self._side is not None else None
            graph = torch.cuda.CUDAGraph()
            try:
                torch.cuda.synchronize()
                with torch.cuda.graph(graph):
                    loss = self.loss_and_grad(cb, zb, sync=True, dropout_key="captured" if key is not None else None, _fused=True)
                G = dict(graph=graph, c=cb, z=zb, loss=loss, sites=dict(self._capture["sites"]))
            except BaseException:
                self.step_count -= 1
                raise
            finally:
                self._capture, self._fused, self._defer_side = None, False, None
            self._graphs[gkey] = G
            while len(self._graphs) > 4:      # every captured shape keeps its activation pool (~6 GB at batch 8): the oldest goes first
                self._graphs.pop(next(iter(self._graphs)))
        # this step's words: AdamW bias corrections and the seed of every dropout site, one host -> device copy ahead of the replay
        import ctypes as C
        bc = (C.c_float * 3)()
        L.check(L.lib().sfmi_adamw_bias_corrections(self.betas[0], self.betas[1], self.step_count, bc), "bias corrections")
        bc[2] = self.lr
        slot = self._words_ring[self._words_next]
        self._words_next = (self._words_next + 1) % len(self._words_ring)
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        w = slot["buf"]
        w[:3] = torch.frombuffer(bytearray(bytes(bc)), dtype=torch.int32)
        for name, idx in G["sites"].items():
            v = _fnv1a32(f"dropout-{key}-{name}")
            w[4 + idx] = v - (1 << 32) if v >= (1 << 31) else v
        self._words.copy_(w, non_blocking=True)
        slot["ev"] = torch.cuda.Event()
        slot["ev"].record()
        G["c"].copy_(c.to(torch.int32), non_blocking=True)
        G["z"].copy_(z.to(torch.int32), non_blocking=True)
        G["graph"].replay()
        self.buckets.end_step()
        self.g.mark_decode_weights_stale()
        return G["loss"].clone()
