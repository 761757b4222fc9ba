// CondTupleGPT (20+4 pre-LN blocks, two tuple heads) KV-cached sampling for gfx950.
//
// Replaces the reference's host-driven loop that re-forwards the WHOLE prefix every step
// (shapeformer.py:54-123 -> mingpt.py:297-310, ~113 TFLOP/sequence) with a KV-cached decode step
// (valid by SURVEY Appendix A11) made of a handful of weight-streaming kernels that read all
// per-row state (lengths, tokens) from device memory, so one captured hipGraph replays every step:
//
//   dgemm     LayerNorm-fused weight-streaming GEMM for M <= 96 rows per launch on f32 MFMA 16x16x4 (mingpt.py:103-111):
//             weights AND activations in MFMA-fragment order (every wave access = 1 KiB contiguous), final
//             outputs (bias / GELU / residual in the epilogue), optional in-kernel deterministic split-K
//   attn      one workgroup per (row, head): KV append, softmax(QK^T/8)V over the row's own cached length
//             (rows are ragged), head-contiguous (B,H,L,64) cache                 (mingpt.py:73-91)
//   rowprep   prefill only: embeddings / residual / LayerNorm over (B*P) rows (mingpt.py:256-286)
//   sample    sampling_masker + filter_sampling_logits + inverse-CDF draw   (representers.py:120-155,
//             common.py:260-299) fused per row: radix-select top-k, bitonic sort of the candidates,
//             top-p cut, counter-hash uniforms (no host RNG, no per-step D2H of (B,4097) logits)
#include "sfmi_common.h"
#include <mutex>
#include <string>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define g_tune g_sfmi_tune   // launch-shape knobs (csrc/capi.hip: sfmi_tune_set)


// ------------------------------------------------------------------------------------------------
// rowprep
// ------------------------------------------------------------------------------------------------
struct RowPrepArgs {
  // EMBED source (mode 0): token tables
  const float *E0, *E1, *Ex, *pos_emb, *cond_pos_emb;
  const int *seq, *len, *Lc;  // (B,Lmax,2), (B), (B)
  const int* nval;            // prefill rows (b,t) are valid for t < nval[b] (NULL: Lc[b]-1)
  const int* extra;           // optional explicit extra index (B,Lmax) (NULL: AR_N rule, representers.py:188-196)
  int* extra_out;             // optional: the extra index actually used, (M) ints (needed by the embedding backward)
  // ACCUM source (mode 1)
  const float* resid_in;   // (M,D)
  const float* part;       // (S,M,D) or null
  const float* bias;       // (D) or null
  const float* Eadd;       // optional: add Eadd[seq[b][t+1][0]] (stage-1 input, mingpt.py:294/309)
  // outputs
  float* resid_out;        // (M,D) or null
  float* xn;               // (M,D) or null (LayerNorm output)
  const float *gamma, *beta;
  int mode, S, M, D, Lmax, P /*0: decode (t = len[b]-1); >0: prefill rows m=(b,t), t<P*/, end0;
  const int* rowoff;       // optional (nB+1): PACKED prefill rows - sequence b owns rows rowoff[b] .. rowoff[b+1]-1 (t = m - rowoff[b])
  int nB;
};

__global__ __launch_bounds__(256) void rowprep_kernel(RowPrepArgs a) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x;
  int b, t;
  if (a.P && a.rowoff) {   // ragged rows packed back to back: no work on padding positions
    int lo = 0, hi = a.nB;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.rowoff[mid] <= m) lo = mid; else hi = mid; }
    b = lo; t = m - a.rowoff[b];
  } else {
    b = a.P ? m / a.P : m;
    if (a.P) { t = m - b * a.P; } else { t = a.len ? a.len[b] - 1 : 0; }
  }
  const int lc = a.Lc ? a.Lc[b] : 0;
  if (a.P) { int tmax = (a.nval ? a.nval[b] : lc - 1) - 1; if (tmax < 0) tmax = 0; if (t > tmax) t = tmax; }  // padded rows: harmless clamp
  const int nq = a.D / 4;
  f32x4 v[4];
  float s = 0.f;
  int pos = 0, val = 0, ext = 0;
  if (a.mode == 0) {
    const int* tk = a.seq + ((long long)b * a.Lmax + t) * 2;
    pos = tk[0]; val = tk[1];
    if (a.extra) {
      ext = a.extra[(long long)b * a.Lmax + t];
    } else if (t < lc) {
      ext = pos;  // representers.py:191 cond token -> own pos
    } else if (pos == a.end0) {
      ext = a.end0;
    } else {      // representers.py:432-442: first cond pos > pos (cond ascending, ends with end token)
      int lo = 0, hi = lc;
      const int* cs = a.seq + (long long)b * a.Lmax * 2;
      while (lo < hi) { int mid = (lo + hi) >> 1; if (cs[2 * mid] > pos) hi = mid; else lo = mid + 1; }
      ext = cs[2 * (lo < lc ? lo : lc - 1)];
    }
  }
  if (a.mode == 0 && a.extra_out && tid == 0) a.extra_out[m] = ext;
  int addrow = -1;
  if (a.mode == 1 && a.Eadd) addrow = a.seq[((long long)b * a.Lmax + t + 1) * 2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = tid + it * 256;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    v[it] = x;
    if (q >= nq) continue;
    if (a.mode == 0) {
      const f32x4 e0 = reinterpret_cast<const f32x4*>(a.E0 + (long long)pos * a.D)[q];
      const f32x4 e1 = reinterpret_cast<const f32x4*>(a.E1 + (long long)val * a.D)[q];
      const f32x4 ex = reinterpret_cast<const f32x4*>(a.Ex + (long long)ext * a.D)[q];
      const float* pe = t < lc ? a.cond_pos_emb + (long long)t * a.D : a.pos_emb + (long long)(t - lc) * a.D;
      x = ((e0 + e1) + ex) + reinterpret_cast<const f32x4*>(pe)[q];  // mingpt.py:285 order
    } else {
      x = reinterpret_cast<const f32x4*>(a.resid_in + (long long)m * a.D)[q];
      if (a.part) {
        f32x4 p = reinterpret_cast<const f32x4*>(a.part + (long long)m * a.D)[q];
        for (int sp = 1; sp < a.S; ++sp) p = p + reinterpret_cast<const f32x4*>(a.part + ((long long)sp * a.M + m) * a.D)[q];
        if (a.bias) p = p + reinterpret_cast<const f32x4*>(a.bias)[q];
        x = x + p;
      }
      if (addrow >= 0) x = x + reinterpret_cast<const f32x4*>(a.Eadd + (long long)addrow * a.D)[q];
    }
    if (a.resid_out) reinterpret_cast<f32x4*>(a.resid_out + (long long)m * a.D)[q] = x;
    v[it] = x;
    s += (x[0] + x[1]) + (x[2] + x[3]);
  }
  if (!a.xn) return;
  // LayerNorm (two-pass, biased variance, eps 1e-5)
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)a.D;
  float qv = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (tid + it * 256 >= nq) continue;
    const f32x4 d = v[it] - mean;
    qv += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
  qv = wave_sum(qv);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = qv;
  __syncthreads();
  const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)a.D + 1e-5f);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = tid + it * 256;
    if (q >= nq) continue;
    const f32x4 g = reinterpret_cast<const f32x4*>(a.gamma)[q], be = reinterpret_cast<const f32x4*>(a.beta)[q];
    reinterpret_cast<f32x4*>(a.xn + (long long)m * a.D)[q] = (v[it] - mean) * rstd * g + be;
  }
}

// ------------------------------------------------------------------------------------------------
// decode GEMM with fused LayerNorm / bias / GELU / residual  (M <= 96 rows, 16x16x4 f32 MFMA)
//
//   plain : out[m][n] = act( sum_k x[m][k] W[n][k] + c2[n] ) (+ resid[m][n])
//   LN    : LayerNorm commutes with the GEMM:  LN(x) W^T = rstd[m] (x W'^T - mean[m] c1[n]) + c2[n]
//           with W' = W diag(gamma), c1[n] = sum_k W'[n][k], c2[n] = sum_k beta[k] W[n][k] + bias[n].
//           Every workgroup streams ALL of x (it owns a 16-column n-tile over the full K), so it derives
//           mean/rstd of each row on the fly - no separate LayerNorm kernel, no split-K partials, outputs
//           are final (deterministic: one workgroup owns each output element).
//   Weights: Wp16 fragment order [N/16][K/16][64][4]; NW waves split K; UN loads per wave in flight.
// ------------------------------------------------------------------------------------------------
// In-situ launch timing (bench.py `roofline`): when a launch is given a `prof` sink {t0, sum of durations, launches} (3 x u64, ticks of
// the constant 100 MHz wall clock) and a finished-workgroup counter, every 64th workgroup records the earliest start with an atomic min
// and the LAST workgroup to finish adds (now - earliest start) - the launch's duration as a kernel trace sees it, minus the dispatch
// ramp - and re-arms the slot.  One sink per chain (launches of one chain never overlap).  prof == NULL: no code runs.
__device__ __forceinline__ void prof_begin(unsigned long long* prof, unsigned wg) {
  if (prof && threadIdx.x == 0 && (wg & 63u) == 0u)
    __hip_atomic_fetch_min(prof, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void prof_end_last(unsigned long long* prof) {   // called by ONE thread of the launch's last workgroup
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __hip_atomic_exchange(prof, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t0 != ~0ull && t1 > t0) {
    __hip_atomic_fetch_add(prof + 1, t1 - t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(prof + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct DGemmArgs {
  const float* x; const float* Wp; const float* c1; const float* c2; const float* resid; float* out;
  int M, N, K, ldo /*row stride when out is row-major (out_packed == 0)*/, ln, act, out_packed;
  float* slab; int* cnt;   // split-K scratch: ceil(M/16)*ceil(N/16)*S*320 floats, ceil(M/16)*ceil(N/16) ints (zeroed once)
  int* pblk; unsigned long long* prof;   // optional in-situ launch timing (prof_begin / prof_end_last)
  int prio;                              // s_setprio level of every wave of the launch (0 = hardware default; knob dgemm_prio)
};

// Decode activations live in MFMA-fragment-packed layout: an (M x N) tensor is stored as
// [ceil(M/16)][N/16][64 lanes][4] with lane = ((n>>2)&3)*16 + (m&15), j = n&3.  That is simultaneously the
// 16x16x4 C/D register layout of the producing GEMM (store index == lane) and the B-operand fragment of the
// consuming GEMM (k == n), so every activation load/store of the decode step is a 1 KiB contiguous wave access.
__device__ __host__ __forceinline__ long long pk_off(int m, int n, int N) {
  return ((((long long)(m >> 4) * (N >> 4) + (n >> 4)) * 64 + ((n >> 2) & 3) * 16 + (m & 15)) << 2) + (n & 3);
}

// 8-byte agent-scope relaxed atomics lower to sc1 (write-through / L1-bypassing) accesses on gfx950: the
// placement-independent hand-off form for small split-K slabs (no release/acquire fences needed).
__device__ __forceinline__ void st_sc1(float* p, f32x4 v) {
  unsigned long long lo = ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]);
  unsigned long long hi = ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
  unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
               __uint_as_float((unsigned)(hi >> 32))};
}

// UN = loads in flight per wave per operand; the host picks UN | steps so the unrolled batches carry NO per-element
// conditions (a runtime select around a load makes hipcc wait vmcnt(0) per element - guide §5 trap 4c).
//
// Operand / instruction policy of dgemm_kernel.  The product instantiates DgProduct only.  tools/ubench/dg_ablation.h supplies
// timing-only ablation policies (operands not loaded, MFMA replaced, statistics or epilogue skipped) for the micro-benchmarks
// without touching this file; nothing here reads the environment or a build-time switch.
struct DgProduct {
  static constexpr bool kStatsOnlyIfLn = false, kNoStats = false, kSkipEpilogue = false;
  // plain loads: the weight stream allocates in the Infinity Cache, so the other interleaved chains re-read a layer's weights from there
  static __device__ __forceinline__ f32x4 wload(const f32x4* p) { return *p; }
  static __device__ __forceinline__ int xidx(int i) { return i; }
  static __device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};
template <int MT, int NW, int UN, int NT = 1, class P = DgProduct>
__global__ __launch_bounds__(64 * NW) void dgemm_kernel(DGemmArgs a) {
  // NT n-tiles per wave (NT = 2: an activation fragment feeds two weight tiles - 5 operand loads per 24 MFMAs instead of 7 for the
  // same 6 accumulator tiles - and the 96-row launch becomes two row groups of MT = 3, whose batch of two k16-steps fits 128 VGPRs,
  // which the 6-row-tile instance does not).  VT = MT * NT "virtual tiles" (row tile j, column tile nn) per wave.
  constexpr int VT = MT * NT;
  __shared__ __attribute__((aligned(16))) float red[NW][VT][4][64];
  __shared__ float st1[NW][MT][16], st2[NW][MT][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, ml = lane & 15;
  const int sp = blockIdx.y, S = gridDim.y;
  // wave priority (the immediate has to be a constant): beside another chain's KV stream the MFMA family then wins the SIMD's issue slots
  if (a.prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.prio == 2) __builtin_amdgcn_s_setprio(2); else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
  prof_begin(a.prof, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  const int ntiles = (a.N + 15) >> 4;
  const int nt0 = blockIdx.x * NT;             // first n-tile of this workgroup (an odd tile count: the last workgroup's second tile is masked)
  // ROW GROUPS (gridDim.z > 1): a launch of more than MT row tiles is cut into groups of MT row tiles; group g = blockIdx.z owns the row
  // tiles t0 .. t0+MT-1 of the fragment-packed operands (row tiles are the outermost dimension of that layout, so a group is a
  // contiguous slab).  Every row keeps the arithmetic of the one-group kernel; the groups of an n-tile stream the same weight
  // slice (the later ones from the XCD's L2 / the Infinity Cache), so one launch serves up to 192 rows per chain with ONE set of
  // launch boundaries and one HBM pass over the weights.
  const int t0 = blockIdx.z * MT;
  const int kslice = a.K / S;
  const int kw = kslice / NW;
  const int k0 = sp * kslice + wave * kw;
  // operand addresses = wave-uniform base (scalar registers) + one 32-bit per-lane offset: the six activation pointers of a
  // 96-row launch would otherwise cost 12 vector registers, the difference between one and two resident workgroups
  const f32x4* wp[NT];
#pragma unroll
  for (int nn = 0; nn < NT; ++nn)
    wp[nn] = reinterpret_cast<const f32x4*>(a.Wp) + ((long long)min(nt0 + nn, ntiles - 1) * (a.K / 16) + k0 / 16) * 64;
  const f32x4* xr[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) xr[j] = reinterpret_cast<const f32x4*>(a.x) + ((long long)(t0 + j) * (a.K / 16) + k0 / 16) * 64;
  const unsigned lo = (unsigned)lane;
  f32x4 acc[MT][NT][2];
  float s1[MT], s2[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) { acc[j][nn][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][nn][1] = acc[j][nn][0]; }
  }
  // epilogue operands are fetched NOW (wave v owns virtual tile v) so that no dependent global round trip is left
  // after the weight stream: the memory system is saturated by then and a late load costs ~1.5 us.
  f32x4 pc1 = {0.f, 0.f, 0.f, 0.f}, pc2 = pc1, pres = pc1;
  const int ej = wave / NT, enn = wave % NT, ent = nt0 + enn;       // (row tile, column tile) this wave finishes first
  const int n_ep = ent * 16 + 4 * q;
  const long long off_ep = a.out_packed ? (((long long)(t0 + ej) * (a.N >> 4) + ent) * 64 + lane) * 4
                                        : (long long)min((t0 + ej) * 16 + ml, a.M - 1) * a.ldo + n_ep;
  if (wave < VT && ent < ntiles && n_ep < a.N) {
    if (a.ln) pc1 = *reinterpret_cast<const f32x4*>(a.c1 + n_ep);
    if (a.c2) pc2 = *reinterpret_cast<const f32x4*>(a.c2 + n_ep);
    if (a.resid) pres = *reinterpret_cast<const f32x4*>(a.resid + off_ep);
  }
  const int steps = kw / 16;
  // software pipeline over batches of UN k16-steps: the loads of batch b+1 are issued BEFORE the MFMAs of batch b
  // (two register sets, statically indexed), and every load of a batch is pinned ahead of the first MFMA that
  // follows (sched_barrier) - hipcc otherwise sinks loads next to their uses and the kernel turns latency-bound.
  auto mfma_step = [&](const f32x4 (&wv)[NT], const f32x4 (&xs)[MT]) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const f32x4 xv = xs[j];
      // LayerNorm statistics (a few VALU ops; computed unconditionally, only used when a.ln)
      if (!P::kNoStats && (!P::kStatsOnlyIfLn || a.ln)) {
        s1[j] += (xv[0] + xv[1]) + (xv[2] + xv[3]);
        s2[j] += (xv[0] * xv[0] + xv[1] * xv[1]) + (xv[2] * xv[2] + xv[3] * xv[3]);
      }
#pragma unroll
      for (int nn = 0; nn < NT; ++nn)
#pragma unroll
        for (int e = 0; e < 4; ++e)   // two independent accumulator chains hide the 40-cycle dependent latency
          acc[j][nn][e & 1] = P::mfma(wv[nn][e], xv[e], acc[j][nn][e & 1]);
    }
  };
  // batches of UN k16-steps: UN*NT weight + UN*MT activation loads in flight, all pinned ahead of the MFMAs
  // (measured alternatives for MT > 1 — two-deep register pipeline, up-front weight preload — were 4-6 % slower in isolation; round 3:
  //  the wave's whole weight slice by LDS-DMA at t = 0 + double-buffered activations, built to make the launch tolerant of the
  //  3-4x memory latency beside a KV stream, was slower alone (1.65 vs 1.50 ms per 80-row chain step) AND beside the stream:
  //  profiles/r03_ar_overlap.md)
  for (int s0 = 0; s0 < steps; s0 += UN) {
    f32x4 w[UN][NT], xb[UN][MT];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
#pragma unroll
      for (int nn = 0; nn < NT; ++nn) w[u][nn] = P::wload(wp[nn] + (s0 + u) * 64 + lo);
#pragma unroll
      for (int j = 0; j < MT; ++j) xb[u][j] = xr[j][P::xidx((s0 + u) * 64) + lo];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UN; ++u) mfma_step(w[u], xb[u]);
  }
  if (P::kSkipEpilogue) {   // ablation policies only: main loop alone (one store per lane keeps the accumulators alive)
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MT; ++j) t = t + acc[j][0][0] + acc[j][0][1];
    if (t[0] == 1.2345f) a.out[tid] = t[1] + s1[0] + s2[0] + pc1[0] + pc2[0] + pres[0];
    return;
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const f32x4 t = acc[j][nn][0] + acc[j][nn][1];
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][j * NT + nn][r][lane] = t[r];
    }
    if (a.ln) {
      float t1 = s1[j], t2 = s2[j];
      t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) { st1[wave][j][ml] = t1; st2[wave][j][ml] = t2; }
    }
  }
  __syncthreads();
  // epilogue: wave w finishes the virtual tiles v = w, w+NW, ...
  for (int v = wave; v < VT; v += NW) {
    const int j = v / NT, nt = nt0 + v % NT;
    if (nt >= ntiles) continue;          // masked second tile of an odd tile count (wave-uniform)
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] += red[w][v][e][lane];
    float t1 = 0.f, t2 = 0.f;
    if (a.ln) {
#pragma unroll
      for (int w = 0; w < NW; ++w) { t1 += st1[w][j][ml]; t2 += st2[w][j][ml]; }
    }
    if (S > 1) {
      // split-K: publish this slice's slab write-through, take a ticket; the last arriver of the (nt, j) tile
      // sums the S slabs in slice order (deterministic) and runs the epilogue.
      const long long tile = (long long)(t0 + j) * ntiles + nt;
      float* slab = a.slab + (tile * S + sp) * 320;          // 256 acc floats + 64 stat floats
      st_sc1(slab + lane * 4, r);
      if (a.ln && q == 0) {
        __hip_atomic_store(slab + 256 + ml, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slab + 272 + ml, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int ticket = 0;
      if (lane == 0) ticket = __hip_atomic_fetch_add(a.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = __shfl(ticket, 0, 64);
      if (ticket != S - 1) continue;
      if (lane == 0) __hip_atomic_store(a.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      r = f32x4{0.f, 0.f, 0.f, 0.f}; t1 = 0.f; t2 = 0.f;
      const float* base = a.slab + tile * S * 320;
      for (int s = 0; s < S; s += 4) {   // four slices in flight per round trip, summed in slice order
        f32x4 tv[4];
        float u1[4], u2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* b = base + min(s + u, S - 1) * 320;
          tv[u] = ld_sc1(b + lane * 4);
          u1[u] = __hip_atomic_load(b + 256 + ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          u2[u] = __hip_atomic_load(b + 272 + ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s + u < S) { r = r + tv[u]; t1 += u1[u]; t2 += u2[u]; }
      }
    }
    const int m = (t0 + j) * 16 + ml;
    const int n = nt * 16 + 4 * q;
    if ((a.out_packed || m < a.M) && n < a.N) {
      // the operands fetched ahead belong to this wave's FIRST virtual tile; later ones (VT > NW only) are fetched here
      f32x4 qc1 = pc1, qc2 = pc2, qres = pres;
      const long long off = a.out_packed ? (((long long)(t0 + j) * (a.N >> 4) + nt) * 64 + lane) * 4 : (long long)m * a.ldo + n;
      if (v != wave) {
        qc1 = a.ln ? *reinterpret_cast<const f32x4*>(a.c1 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        qc2 = a.c2 ? *reinterpret_cast<const f32x4*>(a.c2 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        qres = a.resid ? *reinterpret_cast<const f32x4*>(a.resid + off) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (a.ln) {
        const float mean = t1 / (float)a.K;
        const float var = fmaxf(t2 / (float)a.K - mean * mean, 0.f);
        const float rstd = rsqrtf(var + 1e-5f);
        r = (r - qc1 * mean) * rstd;
      }
      if (a.c2) r = r + qc2;
      if (a.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = 0.5f * r[e] * (1.0f + erff(r[e] * 0.70710678118654752f));
      }
      if (a.resid) r = r + qres;
      *reinterpret_cast<f32x4*>(a.out + off) = r;
    }
  }
  if (a.prof && tid == 0 &&
      __hip_atomic_fetch_add(a.pblk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)(gridDim.x * gridDim.y * gridDim.z) - 1) {
    __hip_atomic_store(a.pblk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prof_end_last(a.prof);
  }
}

// ------------------------------------------------------------------------------------------------
// decode attention: one work item per (row, head), NWV waves per workgroup; head dim HD <= 64, multiple of 4.
// KV cache layout is (B, H, Lmax, HD): one (row, head)'s keys are CONTIGUOUS, so a wave-instruction reads
// 4 keys x 256 B = 1 KiB coalesced; 16 lanes share a key (float4 each), U independent loads are in
// flight per lane (unrolled), all waves of the workgroup stream disjoint keys.  A lane meets its keys in
// ascending order whatever U is, so U changes the loads in flight, never the result.
// The launch is either one workgroup per item (grid = B*H, item = row + B*head: the block -> XCD map of a
// (B, H) grid) or PERSISTENT: a fixed number of workgroups stride over the items, so that the whole grid is
// dispatched in one round and workgroups of OTHER kernels (the decode GEMMs of the other chains) can be
// placed beside it for its whole duration (profiles/r03_ar_overlap.md).
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* qkv;      // fragment-packed (M x 3D) q|k|v of the fused LN+QKV GEMM (bias included)
  float* Kc; float* Vc;  // (B,H,Lmax,HD)
  const int* len; float* y /*fragment-packed (M x D)*/;
  const int* shared_len;
  int B, H, D, Lmax, HD; float scale;
  int* sem;      // optional turnstile words {next ticket, finished launches, gate time-outs}: the LAST workgroup of a launch bumps sem[1]
  int* blk;      // this chain's finished-workgroup counter (re-armed by the last workgroup)
  unsigned long long* prof;   // optional in-situ launch timing sink of this chain (prof_begin / prof_end_last; needs blk)
};
template <int NWV>
struct AttnLds {
  __attribute__((aligned(16))) float qs[64];
  __attribute__((aligned(16))) float kn[64];
  __attribute__((aligned(16))) float vn[64];
  __attribute__((aligned(16))) float sc[1024];
  float red[2 * NWV];
  __attribute__((aligned(16))) float yacc[NWV][64];
};

// (Round 4 measured a "small grid" form for 16-32 rows - 8 loads in flight per lane and the first batch of values requested together
// with the keys, one round trip instead of two - and dropped it: 0.47 against 0.39 ms per step of attention at 16 rows, the extra loads
// queue in front of the keys the scores wait for.  profiles/r04_b16_experiments.md)
// SH: the launch has a shared prefix (a.shared_len != NULL).  The plain instance keeps ONE wave-uniform base per cache (scalar
// registers, scalar-base loads); the per-lane choice between row 0's cache and the row's own costs a compare and two selects on a
// 64-bit address per load, and the product instance (64 registers, 8 waves per SIMD) has no slack for them: 893 -> 748 vector
// instructions, 79 -> 43 scalar-spill reads, 7.36 -> 7.27 ms per 384-row step (bench 84.1 -> 86.3 shapes/s); with the unconditional
// loads below 653 instructions and 7.08 ms (87.8).  A further
// specialisation on head dimension 64 (no column test) drops to 691 instructions but spills 20 bytes of vector registers: not taken.
template <int NWV, int U, bool SH>
__device__ __forceinline__ void attn_decode_item(AttnLds<NWV>& s, const AttnArgs& a, const int b, const int h) {
  constexpr int KB = NWV * 4 * U;   // keys per batch of loads
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, HD = a.HD, D = a.D, Lmax = a.Lmax;
  const int t = __builtin_amdgcn_readfirstlane(a.len[b] - 1);  // position being processed
  const int nq4 = HD / 4;
  float* Kb = a.Kc + ((long long)b * H + h) * Lmax * HD;
  float* Vb = a.Vc + ((long long)b * H + h) * Lmax * HD;
  // shared prefix (sample_n copies of ONE condition, shapeformer.py:222-260): keys / values of positions < shared_len[0] were
  // written once, by row 0's prefill, and every row reads them from row 0's cache (one HBM read, L2 / Infinity-Cache hits
  // for the other rows); a row's own cache holds its tail only
  const int nshared = (SH && a.shared_len) ? a.shared_len[0] : 0;
  const float* Kb0 = SH ? a.Kc + (long long)h * Lmax * HD : Kb;
  const float* Vb0 = SH ? a.Vc + (long long)h * Lmax * HD : Vb;
  const int c4 = lane & 15, kk = lane >> 4;
  const bool cok = c4 < nq4;
  const float* safe = a.qkv + pk_off(b, 0, 3 * D);      // 16-byte aligned, finite: this row's packed q of the current step
  // the first batch of keys is requested BEFORE the q/k/v hand-off barrier (the loads only need `t`): the HBM latency of the
  // first batch overlaps the LDS round trip instead of following it
  auto load_k = [&](int i0, f32x4 (&kf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      if (SH) {
        kf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < t && cok) kf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((i < nshared ? Kb0 : Kb) + (long long)i * HD + 4 * c4));
      } else {
        // plain instance: an UNCONDITIONAL load (no exec-mask region, no zero fill per load).  A lane without a cached key (i >= t,
        // or a column beyond the head) reads the step's own q row instead - always finite - and its score is discarded below
        const float* p = (i < t && cok) ? Kb + (i * HD + 4 * c4) : safe;
        kf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
      }
    }
  };
  auto load_v = [&](int i0, f32x4 (&vf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      if (SH) {
        vf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < t && cok) vf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((i < nshared ? Vb0 : Vb) + (long long)i * HD + 4 * c4));
      } else {
        // (a finite stand-in value times a probability of exactly 0 adds +-0 to the accumulator: the sums are unchanged bit for bit)
        const float* p = (i < t && cok) ? Vb + (i * HD + 4 * c4) : safe;
        vf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
      }
    }
  };
  f32x4 kf0[U], vf0[U];
  load_k(0, kf0);
  if (tid < HD) {
    const float q = a.qkv[pk_off(b, h * HD + tid, 3 * D)];
    const float k = a.qkv[pk_off(b, D + h * HD + tid, 3 * D)];
    const float v = a.qkv[pk_off(b, 2 * D + h * HD + tid, 3 * D)];
    s.qs[tid] = q * a.scale; s.kn[tid] = k; s.vn[tid] = v;
    Kb[(long long)t * HD + tid] = k;
    Vb[(long long)t * HD + tid] = v;
  }
  __syncthreads();
  f32x4 qf = {0.f, 0.f, 0.f, 0.f};
  if (cok) qf = *reinterpret_cast<const f32x4*>(s.qs + 4 * c4);
  float lmax = -INFINITY;
  auto score = [&](int i0, f32x4 (&kf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      if (i == t && cok) kf[u] = *reinterpret_cast<const f32x4*>(s.kn + 4 * c4);     // the new token's key comes from LDS
      float d = (qf[0] * kf[u][0] + qf[1] * kf[u][1]) + (qf[2] * kf[u][2] + qf[3] * kf[u][3]);
      d = row16_sum(d);      // over the 16 lanes that share the key (DPP: no LDS round trips)
      if (i <= t) {
        if (c4 == 0) s.sc[i] = d;
        lmax = fmaxf(lmax, d);
      }
    }
  };
  score(0, kf0);
  for (int i0 = KB; i0 <= t; i0 += KB) {
    f32x4 kf[U];
    load_k(i0, kf);
    score(i0, kf);
  }
  // same for the values: the first batch is requested before the softmax barrier
  load_v(0, vf0);
  lmax = wave_max(lmax);
  if (lane == 0) s.red[wave] = lmax;
  __syncthreads();
  float gmax = s.red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) gmax = fmaxf(gmax, s.red[w]);
  // y = sum_i p_i V[i], p_i = exp(s_i - gmax); the 1/sum is applied at the end
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float ls = 0.f;
  auto accum = [&](int i0, f32x4 (&vf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      if (i == t && cok) vf[u] = *reinterpret_cast<const f32x4*>(s.vn + 4 * c4);
      const float pr = i <= t ? __expf(s.sc[i] - gmax) : 0.f;
      acc = acc + vf[u] * pr;
      ls += pr;
    }
  };
  accum(0, vf0);
  for (int i0 = KB; i0 <= t; i0 += KB) {
    f32x4 vf[U];
    load_v(i0, vf);
    accum(i0, vf);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc[e] += __shfl_xor(acc[e], 16, 64); acc[e] += __shfl_xor(acc[e], 32, 64); }
  ls += __shfl_xor(ls, 16, 64); ls += __shfl_xor(ls, 32, 64);  // every c4 lane holds the wave's sum over its kk keys
  if (kk == 0 && cok) *reinterpret_cast<f32x4*>(&s.yacc[wave][4 * c4]) = acc;
  if (lane == 0) s.red[NWV + wave] = ls;
  __syncthreads();
  if (tid < HD) {
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) { o += s.yacc[w][tid]; l += s.red[NWV + w]; }
    a.y[pk_off(b, h * HD + tid, D)] = o / l;   // fragment-packed (M x D)
  }
}

template <int NWV, int U, bool SH>
__global__ __launch_bounds__(64 * NWV, U <= 4 ? 8 : U <= 8 ? 4 : 2) void attn_decode_kernel(AttnArgs a) {   // U <= 4: <= 64 VGPRs (8 waves per SIMD), U = 8: <= 128, U = 16: <= 256
  __shared__ AttnLds<NWV> s;
  const int nitems = a.B * a.H;
  prof_begin(a.prof, blockIdx.x);
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int h = __builtin_amdgcn_readfirstlane(it / a.B);   // wave-uniform: keeps the cache bases in scalar registers
    attn_decode_item<NWV, U, SH>(s, a, __builtin_amdgcn_readfirstlane(it - h * a.B), h);
    if (it + (int)gridDim.x < nitems) __syncthreads();   // the next item rewrites the hand-off tiles
  }
  if ((a.sem || a.prof) && threadIdx.x == 0) {     // turnstile release: the launch's last workgroup to finish admits the next KV stream
    if (__hip_atomic_fetch_add(a.blk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_store(a.blk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.sem) __hip_atomic_fetch_add(a.sem + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.prof) prof_end_last(a.prof);
    }
  }
}

// Turnstile in front of a decode-attention launch (one wavefront): takes a ticket and waits until fewer than `lanes` of the
// earlier tickets are still streaming.  The decode chains are independent hipGraphs on separate hardware queues; without the
// turnstile two or three of them are in their attention phase at any time, those launches together take every wave slot and
// vector register of every CU, and the other chains' GEMM workgroups cannot be placed until one of them drains: HBM-bound and
// MFMA-bound phases then time-share the chip instead of overlapping (profiles/r03_ar_overlap.md).  With at most `lanes` KV
// streams resident - each sized to saturate HBM with half of a CU's registers - a GEMM workgroup always fits beside them.
// Carries no data (pure scheduling): a wrong order can only cost time.  A wait longer than 20 ms gives up, counts in sem[2] and
// disarms the turnstile until the caller re-arms it (zeroes sem).
__global__ __launch_bounds__(64) void attn_gate_kernel(int* sem, int lanes) {
  if (threadIdx.x != 0) return;
  const int my = __hip_atomic_fetch_add(sem, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // a gate that ever timed out (two chains on one hardware queue after all: the gate would be holding up the very launch it waits
  // for) switches the turnstile off for the rest of the run - the chains then run ungated, as in round 2, instead of paying 20 ms each
  if (__hip_atomic_load(sem + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
  while (__hip_atomic_load(sem + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + lanes <= my) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) {
      __hip_atomic_fetch_add(sem + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// causal self-attention over a prefix on the MATRIX cores (prefill of the condition, teacher-forced forward, training
// forward; mingpt.py:73-91): rows (b,t<P) of qkv (rectangle or packed, see sfmi_gpt_attn_prefill_f32), also fills the KV
// caches; grid (B, H, ceil(P/64)), 4 waves, wave w owns query
// rows q0+16w .. +15 and walks the key blocks 0 .. q0 (64 keys each, staged once per workgroup in LDS).
//   S (16 x 64)  = Q K^T   : 4 key tiles x 16 k-steps of v_mfma_f32_16x16x4_f32, Q fragments live in 16 registers
//   online softmax on the C/D layout (row 4(l>>4)+j lives in register j of lanes with the same l>>4: 16-lane reductions)
//   O (16 x 64) += P V     : P goes through a per-wave LDS tile to become an A operand; 4 d tiles x 16 k-steps
// LDS row strides HD + 4 (K, V) and 68 (P), rows in operand order (below).
// ------------------------------------------------------------------------------------------------
// Operand-order LDS rows (round 4): inside a row the elements are PERMUTED so that what one lane feeds to consecutive MFMAs is
// contiguous and comes in with ds_read_b128 instead of four ds_read_b32 (144 -> 36 LDS read instructions per 64-key block and wave):
//   K row (one key):   dim d   at (d % 4) * KK + d / 4       (lane lq reads its KK k-steps 4 kk + lq in a row)
//   V row (one key):   dim d   at (d % 16) * DT + d / 16     (lane lr reads its DT output tiles 16 dt + lr in a row)
//   P row (one query): key k   at (k % 4) * 16 + k / 4       (lane lq reads its 16 k-steps 4 kk + lq in a row)
// The MFMA operands and their order are those of the linear layout: results are bit-identical.
constexpr int AP_PS = 68;   // P tile row stride (16 rows x 64 keys)
template <int HD>           // head dim 16 / 32 / 64
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ Kc,
                                                                float* __restrict__ Vc, const int* __restrict__ nval,
                                                                float* __restrict__ y, int P, int D, int Lmax, float scale,
                                                                const int* __restrict__ rowoff, float drop_p, unsigned drop_seed,
                                                                float* __restrict__ lse, const unsigned* __restrict__ drop_seed_dev) {
  if (drop_seed_dev) drop_seed = *drop_seed_dev;      // training step captured in a hipGraph: the seed of this site lives in device memory
  // row strides HD + 4: 16-byte aligned rows whose ds_read_b128 of 16 consecutive rows hit distinct banks; 51 KB of LDS per workgroup
  // at HD = 64 -> three resident workgroups per CU (a workgroup walks only 1-4 key blocks: its prologue latency needs company)
  constexpr int AP_KS = HD + 4, AP_VS = HD + 4, KK = HD / 4, DT = HD / 16;
  __shared__ __attribute__((aligned(16))) float Ks[64 * AP_KS], Vs[64 * AP_VS], Ps[4][16 * AP_PS];
  const int b = blockIdx.x, h = blockIdx.y, qb = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(P, max(nval[b], 0));
  const long long base = rowoff ? rowoff[b] : (long long)b * P;
  const int q0 = qb * 64;
  if (q0 >= n) return;
  const int lr = lane & 15, lq = lane >> 4;
  // Q fragments (A operand: row lr, k = 4 kk + lq), pre-scaled
  float qf[KK];
  {
    const int tq = min(q0 + 16 * wave + lr, n - 1);
    const float* qp = qkv + (base + tq) * 3 * D + h * HD + lq;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = qp[4 * kk] * scale;
  }
  float mrun[4], lrun[4];
  f32x4 o[DT];
#pragma unroll
  for (int j = 0; j < 4; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.f; }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kend = min(n, q0 + 64);
  float* Pw = Ps[wave];
  // K / V blocks go global -> registers -> LDS; the loads of block k0 + 64 are issued right after block k0 has been handed to LDS, so
  // their latency runs under the 128 MFMAs of block k0 (a workgroup walks only 1 - 4 blocks at the bench's condition lengths: an
  // exposed load per block was most of its time).
  constexpr int NIT = (64 * KK + 255) / 256;
  f32x4 kreg[NIT], vreg[NIT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int r = i / KK, c = i % KK;
      if (i < 64 * KK) {
        const int tk = min(k0 + r, n - 1);
        const float* src = qkv + (base + tk) * 3 * D + h * HD + 4 * c;
        kreg[it] = *reinterpret_cast<const f32x4*>(src + D);
        vreg[it] = *reinterpret_cast<const f32x4*>(src + 2 * D);
      }
    }
  };
  gload(0);
  const bool wave_active = q0 + 16 * wave < n;     // a wave whose 16 query rows are all padding only helps with the staging
  for (int k0 = 0; k0 < kend; k0 += 64) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int r = i / KK, c = i % KK;
      if (i < 64 * KK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // dims 4c .. 4c+3 of key row r, scattered into operand order
          Ks[r * AP_KS + e * KK + c] = kreg[it][e];
          Vs[r * AP_VS + ((4 * c + e) & 15) * DT + (c >> 2)] = vreg[it][e];
        }
        if (k0 == q0 && k0 + r < n) {  // this block owns these cache rows
          const long long co = (((long long)b * gridDim.y + h) * Lmax + k0 + r) * HD + 4 * c;  // (B,H,Lmax,HD)
          *reinterpret_cast<f32x4*>(Kc + co) = kreg[it];
          *reinterpret_cast<f32x4*>(Vc + co) = vreg[it];
        }
      }
    }
    __syncthreads();
    if (k0 + 64 < kend) gload(k0 + 64);
    if (!wave_active) continue;
    // key tiles this wave needs from the block: all four below the diagonal; on the diagonal block (k0 == q0) only tiles 0 .. wave
    // (tile t > wave holds keys > every query row of the wave: fully masked, p == 0 exactly, so skipping it changes no bit); never
    // tiles that start at or beyond the sequence end
    const int tmax = min(k0 == q0 ? wave : 3, (kend - k0 - 1) >> 4);
    // S = Q K^T
    f32x4 sacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t > tmax) continue;
      const f32x4* kp = reinterpret_cast<const f32x4*>(&Ks[(16 * t + lr) * AP_KS + lq * KK]);
#pragma unroll
      for (int k4 = 0; k4 < KK / 4; ++k4) {
        const f32x4 kv = kp[k4];
#pragma unroll
        for (int e = 0; e < 4; ++e) sacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[4 * k4 + e], kv[e], sacc[t], 0, 0, 0);
      }
    }
    // causal / length mask, online softmax per row (row of register j: q0 + 16 wave + 4 lq + j)
    float corr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qrow = q0 + 16 * wave + 4 * lq + j;
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t > tmax) continue;
        const int key = k0 + 16 * t + lr;
        if (key > qrow || key >= n) sacc[t][j] = -INFINITY;
        mx = fmaxf(mx, sacc[t][j]);
      }
      mx = row16_max(mx);
      const float mnew = fmaxf(mrun[j], mx);
      const float ms = mnew == -INFINITY ? 0.f : mnew;        // fully masked so far: keep everything at zero
      corr[j] = __expf(mrun[j] - ms);
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t > tmax) continue;
        float p = __expf(sacc[t][j] - ms);
        ps += p;                       // the softmax denominator is the undropped sum (att = softmax; att = attn_drop(att))
        if (drop_p > 0.f)              // training: element (b, h, query, key) of the (B,H,P,P) attention-probability tensor
          p *= sfmi_dropout_mul(drop_seed, (unsigned)(((b * gridDim.y + h) * P + qrow) * P + k0 + 16 * t + lr), drop_p, 1.0f / (1.0f - drop_p));
        Pw[(4 * lq + j) * AP_PS + (lr & 3) * 16 + 4 * t + (lr >> 2)] = p;     // key 16 t + lr in operand order
      }
      ps = row16_sum(ps);
      lrun[j] = lrun[j] * corr[j] + ps;
      mrun[j] = mnew;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] *= corr[j];
    __builtin_amdgcn_wave_barrier();   // P tile written and read by this wave only; LDS ops of a wave execute in order
    // O += P V over the keys of the tiles in use
    const f32x4* pp = reinterpret_cast<const f32x4*>(&Pw[lr * AP_PS + lq * 16]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t > tmax) continue;
      const f32x4 pa = pp[t];               // P[query lr][keys 4 (4t + k4) + lq], k4 = 0..3
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int kk = 4 * t + k4;
        const float* vp = &Vs[(4 * kk + lq) * AP_VS + lr * DT];
        float vv[DT];
        if (DT == 4) { const f32x4 v4 = *reinterpret_cast<const f32x4*>(vp); vv[0] = v4[0]; vv[1] = v4[1]; vv[DT > 2 ? 2 : 0] = v4[2]; vv[DT > 3 ? 3 : 0] = v4[3]; }
        else {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) vv[dt] = vp[dt];
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[k4], vv[dt], o[dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tq = q0 + 16 * wave + 4 * lq + j;
    if (tq >= n) continue;
    const float inv = 1.0f / lrun[j];
    float* yp = y + (base + tq) * D + h * HD + lr;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) yp[16 * dt] = o[dt][j] * inv;
    // training forward: the row's log-sum-exp of the scaled scores, (B,H,P) - the backward pass starts from it instead of recomputing Q K^T
    if (lse && lr == 0) lse[((long long)b * gridDim.y + h) * P + tq] = mrun[j] + __logf(lrun[j]);
  }
}

// ------------------------------------------------------------------------------------------------
// fused sampler: one workgroup per row
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned fkey_u(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float sf_uniform(unsigned seed, unsigned idx) { return sfmi_hash_unit(seed, idx); }  // == weights.hash_unit

struct SampleArgs {
  const float* part;  // (S,M,ldv) logits (S partial slabs summed in order; heads have no bias)
  int* seq; int* len; const int* Lc;
  float* logp;        // (B,max_steps,2) log-prob of the drawn token under the masked logits, or null
  float* hist;        // (B,max_steps,V) masked logits history for this tuple element, or null
  const int* force;   // (B,max_steps,2) teacher-forced tokens (parity tests), or null
  // fused tails (decode step): tuple 0 -> resid[b] += E0[pos']   (stage-1 input, mingpt.py:294/309)
  //                            tuple 1 -> resid[b]  = embedding of the token just completed (mingpt.py:256-286),
  //                                       i.e. the next step's stage-0 input
  float* resid; const float *E0, *E1, *Ex, *pos_emb;
  int D;
  int S, M, V, ldv, Lmax, tuple_i, end0, end1, top_k, greedy_row0, mask_invalid, mask_completion, max_steps, advance;
  float top_p, temperature;
  unsigned seed;
  const unsigned* seed_dev;     // optional: the seed lives in device memory (a captured hipGraph is reused across seeds)
  int row_offset, rows_total;   // micro-batching: global row = row_offset + blockIdx.x of rows_total (uniform stream, greedy row 0)
  int step_offset;              // tokens generated BEFORE this call (a non-empty z_indices, shapeformer.py:60-70): the step counter
                                // of masker / history / log-prob / uniforms restarts at 0 at the first new token, as the reference's does
  float* mask_out;              // mask-only mode (sfmi_gpt_mask_logits_f32): masked logits -> mask_out[b*V + v], nothing else happens
};

#define SMP_MAXC 512
#define SMP_BIG 8192
#define SMP_NPOS 1040   // token positions of one row staged in LDS (block_size + 1 <= 1025 on every shipped config)
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  __shared__ float lg[4352];
  __shared__ unsigned hist[256];
  __shared__ __attribute__((aligned(16))) float cval_s[SMP_MAXC], cexp_s[SMP_MAXC];
  __shared__ int cidx_s[SMP_MAXC];
  __shared__ int spos[SMP_NPOS];
  __shared__ float redf[8];
  __shared__ int redi[8];
  __shared__ unsigned s_prefix, s_need;
  __shared__ int s_cnt, s_choice, s_keep;
  __shared__ float s_tot;
  extern __shared__ __attribute__((aligned(16))) float dyn_lds[];  // big path: [NS] values, [NS] indices, [NS] exps
  // candidate buffer: 512 static entries normally; the whole (padded) vocabulary when top_k is 0 or > 512
  const bool big = a.top_k <= 0 || a.top_k > SMP_MAXC;
  const int NSMAX = big ? SMP_BIG : SMP_MAXC;
  float* cval = big ? dyn_lds : cval_s;
  int* cidx = big ? reinterpret_cast<int*>(dyn_lds + SMP_BIG) : cidx_s;
  float* cexp = big ? dyn_lds + 2 * SMP_BIG : cexp_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Everything the row needs from HBM is requested in ONE round trip: the logits (registers), the row's state words, and the
  // POSITION column of its token row (LDS, when it fits: Lmax <= SMP_NPOS) - the masks' binary searches over the condition and the
  // last / current positions are then LDS reads instead of chains of 8-10 dependent global loads (~1 us each: the sampler sits on
  // the decode step's critical path twice per step).  Same values, same arithmetic.
  const int* row = a.seq + (long long)b * a.Lmax * 2;
  const bool staged = a.Lmax <= SMP_NPOS;
  float xr[17];       // V <= 4352 = 17 x 256
#pragma unroll
  for (int r = 0; r < 17; ++r) {
    const int v = tid + 256 * r;
    xr[r] = v < a.V ? a.part[(long long)b * a.ldv + v] : 0.f;
  }
  if (staged)
    for (int i = tid; i < a.Lmax; i += 256) spos[i] = row[2 * i];
  const int L = a.len[b], lc = a.Lc[b];
  __syncthreads();
  auto pos_at = [&](int i) { return staged ? spos[i] : row[2 * i]; };
  const int j = L - lc - a.step_offset;  // step index of this row (the reference's loop counter, shapeformer.py:71)
  const int last_pos = pos_at(L - 1);
  const int cur_pos = pos_at(L);  // valid for tuple 1 (pos just sampled)
  // this draw's uniform (counter hash of (step, tuple, global row)): formed here so that the seed's load is off the serial tail
  const float u_draw = sf_uniform(a.seed_dev ? *a.seed_dev : a.seed, (unsigned)((j * 2 + a.tuple_i) * a.rows_total + a.row_offset + b));
  // next cond position > last_pos (representers.py:141-150), cond list + [end0+1]
  int next_cond = a.end0 + 1;
  if (a.tuple_i == 0 && a.mask_completion) {
    int lo = 0, hi = lc;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (pos_at(mid) > last_pos) hi = mid; else lo = mid + 1; }
    if (lo < lc) next_cond = pos_at(lo);
  }
  // ---- masked logits (sampling_masker) ------------------------------------------------------
  float lmax = -INFINITY;
  int amax = 0;
#pragma unroll
  for (int r = 0; r < 17; ++r) {
    const int v = tid + 256 * r;
    if (v >= a.V) continue;
    float x = xr[r];
    for (int s = 1; s < a.S; ++s) x += a.part[((long long)s * a.M + b) * a.ldv + v];
    if (a.tuple_i == 1) {
      if (cur_pos == a.end0) x = (v == a.end1) ? 1.0f : -INFINITY;
    } else {
      if (a.mask_invalid && j > 0 && v <= last_pos && v != a.end0) x = -INFINITY;
      if (a.mask_completion && v > next_cond) x = -INFINITY;
    }
    lg[v] = x;
    if (a.mask_out) a.mask_out[(long long)b * a.V + v] = x;
    if (a.hist) a.hist[((long long)b * a.max_steps + j) * a.V + v] = x;
    if (x > lmax) { lmax = x; amax = v; }
  }
  if (a.mask_out) return;   // mask-only mode (uniform per launch)
  // block argmax (lowest index on ties) + logsumexp
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(lmax, o, 64);
    const int oi = __shfl_xor(amax, o, 64);
    if (om > lmax || (om == lmax && oi < amax)) { lmax = om; amax = oi; }
  }
  if (lane == 0) { redf[wave] = lmax; redi[wave] = amax; }
  __syncthreads();
  float gmax = redf[0];
  int gidx = redi[0];
  for (int w = 1; w < 4; ++w)
    if (redf[w] > gmax || (redf[w] == gmax && redi[w] < gidx)) { gmax = redf[w]; gidx = redi[w]; }
  float se = 0.f;
  for (int v = tid; v < a.V; v += 256) se += __expf(lg[v] - gmax);
  se = wave_sum(se);
  if (lane == 0) redf[4 + wave] = se;
  __syncthreads();
  const float lse = gmax + __logf((redf[4] + redf[5]) + (redf[6] + redf[7]));

  int choice = gidx;
  const bool greedy = (a.greedy_row0 && a.row_offset + b == 0) || a.top_k == 1 || a.force != nullptr;
  if (!greedy) {
    // ---- top-k threshold by MSB-first radix select on order-preserving keys of lg/T ----------
    const float invT = 1.0f / a.temperature;
    const int k = a.top_k > 0 ? min(a.top_k, a.V) : a.V;
    if (tid == 0) { s_prefix = 0u; s_need = (unsigned)k; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      hist[tid] = 0u;
      __syncthreads();
      const unsigned prefix = s_prefix;
      const unsigned pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int v = tid; v < a.V; v += 256) {
        const unsigned key = fkey_u(lg[v] * invT);
        if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (wave == 0) {  // descending scan over the 256 bins by one wave (4 bins per lane)
        const unsigned need = s_need;
        const int b0 = 255 - 4 * lane;
        const unsigned h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
        const unsigned tot = (h0 + h1) + (h2 + h3);
        unsigned incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        const unsigned excl = incl - tot;
        if (need > excl && need <= incl) {
          unsigned c = excl; int bin = b0;
          if (c + h0 < need) { c += h0; bin = b0 - 1; if (c + h1 < need) { c += h1; bin = b0 - 2; if (c + h2 < need) { c += h2; bin = b0 - 3; } } }
          s_need = need - c;
          s_prefix = prefix | ((unsigned)bin << shift);
        }
      }
      __syncthreads();
    }
    const unsigned kth = s_prefix;  // key of the k-th largest value
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int v = tid; v < a.V; v += 256) {
      const float x = lg[v] * invT;
      if (x > -INFINITY && fkey_u(x) >= kth) {
        const int slot = atomicAdd(&s_cnt, 1);
        if (slot < NSMAX) { cval[slot] = x; cidx[slot] = v; }
      }
    }
    __syncthreads();
    const int C = min(s_cnt, NSMAX);
    int NS = 2;
    while (NS < C) NS <<= 1;
    for (int i = tid; i < NS; i += 256)
      if (i >= C) { cval[i] = -INFINITY; cidx[i] = 0x7fffffff; }
    __syncthreads();
    // order the candidates (value desc, index asc).  Up to 512 of them (top_k <= 512): RANK sort - every candidate counts the
    // candidates that precede it (all lanes read the same LDS word per step: broadcast, conflict-free) and is written to its
    // rank: one pass and two barriers instead of the ~30-45 barrier-separated passes of a bitonic network (14 of the kernel's
    // ~40 us).  The order is total (indices are distinct), so the result is exactly the sorted sequence.  Larger candidate
    // sets (top_k = 0 / > 512, the whole vocabulary) keep the bitonic network.
    if (C <= SMP_MAXC && !big) {
      float mv[2]; int mi[2], rk[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = tid + 256 * e;
        mv[e] = i < C ? cval[i] : -INFINITY; mi[e] = i < C ? cidx[i] : 0x7fffffff; rk[e] = 0;
      }
      for (int j = 0; j < C; ++j) {
        const float vj = cval[j];
        const int ij = cidx[j];
#pragma unroll
        for (int e = 0; e < 2; ++e) rk[e] += (vj > mv[e] || (vj == mv[e] && ij < mi[e])) ? 1 : 0;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (tid + 256 * e < C) { cval[rk[e]] = mv[e]; cidx[rk[e]] = mi[e]; }
      __syncthreads();
    } else {
    // bitonic sort of NS entries (value desc, index asc)
    for (int sz = 2; sz <= NS; sz <<= 1)
      for (int st = sz >> 1; st > 0; st >>= 1) {
        for (int pr = tid; pr < NS / 2; pr += 256) {
          const int i = ((pr / st) * 2 * st) + (pr % st), p = i + st;
          const bool up = (i & sz) == 0;
          const float va = cval[i], vb = cval[p];
          const int ia = cidx[i], ib = cidx[p];
          const bool a_first = va > vb || (va == vb && ia < ib);
          if (a_first != up) { cval[i] = vb; cval[p] = va; cidx[i] = ib; cidx[p] = ia; }
        }
        __syncthreads();
      }
    }
    // exps in parallel; the order-sensitive running sums stay sequential (oracle convention), but only ONE sequential pass over
    // all candidates is left: it also records the running sums (the candidate values are no longer needed, so they go to cval[]).
    // The second sum of the old form (over the kept prefix) IS one of those running sums, the inverse-CDF scan is a search for the
    // first running sum above the threshold (they are non-decreasing: done by all threads), and the top-p quotients are formed in
    // parallel before their sequential accumulation - which stops at the first prefix above top_p, i.e. early.
    const float m0 = cval[0];
    for (int i = tid; i < C; i += 256) cexp[i] = __expf(cval[i] - m0);
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;          // strictly sequential adds (the oracle's order); 16 elements per LDS round trip
      int i = 0;
      for (; i + 16 <= C; i += 16) {
        f32x4 e[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) e[q4] = *reinterpret_cast<const f32x4*>(cexp + i + 4 * q4);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          f32x4 pr;
          tot += e[q4][0]; pr[0] = tot; tot += e[q4][1]; pr[1] = tot; tot += e[q4][2]; pr[2] = tot; tot += e[q4][3]; pr[3] = tot;
          *reinterpret_cast<f32x4*>(cval + i + 4 * q4) = pr;
        }
      }
      for (; i < C; ++i) { tot += cexp[i]; cval[i] = tot; }
      s_tot = tot;
    }
    __syncthreads();
    const float tot = s_tot;
    if (a.top_p > 0.f) {
      for (int i = tid; i < C; i += 256) cexp[i] = cexp[i] / tot;      // softmax probabilities of the sorted candidates
      __syncthreads();
    }
    if (tid == 0) {
      // top-p (common.py:271-284): drop sorted i>=1 where cumsum(softmax)[i-1] > p
      int keep = C;
      if (a.top_p > 0.f) {
        float cum = 0.f;
        keep = 1;
        for (int i = 0; i + 1 < C; ++i) {
          cum += cexp[i];
          if (cum > a.top_p) break;
          keep = i + 2;
        }
      }
      // inverse-CDF draw (oracle/tokens_oracle.py:sample_filtered convention): first i < keep whose running sum exceeds u * sum(kept)
      s_keep = keep;
      s_tot = u_draw * cval[keep - 1];
      s_choice = keep - 1;
    }
    __syncthreads();
    {
      const int keep = s_keep;
      const float thr = s_tot;
      int first = 0x7fffffff;
      for (int i = tid; i < keep; i += 256)
        if (cval[i] > thr) { first = i; break; }       // a thread's indices ascend: its first hit is its smallest
      for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
      if (lane == 0 && first != 0x7fffffff) atomicMin(&s_choice, first);
    }
    __syncthreads();
    if (tid == 0) s_choice = cidx[s_choice];
    __syncthreads();
    choice = s_choice;
  }
  if (a.force && j < a.max_steps) choice = a.force[((long long)b * a.max_steps + j) * 2 + a.tuple_i];
  if (tid == 0) {
    a.seq[((long long)b * a.Lmax + L) * 2 + a.tuple_i] = choice;
    if (a.logp && j < a.max_steps) a.logp[((long long)b * a.max_steps + j) * 2 + a.tuple_i] = lg[choice] - lse;
    if (a.advance) a.len[b] = L + 1;
  }
  // ---- fused tails --------------------------------------------------------------------------
  if (a.resid) {
    const int nq = a.D / 4;
    if (a.tuple_i == 0) {
      const f32x4* e0 = reinterpret_cast<const f32x4*>(a.E0 + (long long)choice * a.D);
      for (int qd = tid; qd < nq; qd += 256) {
        f32x4* rr = reinterpret_cast<f32x4*>(a.resid + pk_off(b, 4 * qd, a.D));
        *rr = *rr + e0[qd];
      }
    } else {
      const int pos = cur_pos, val = choice, t = L;  // the token at position L is now complete
      int ext;
      if (pos == a.end0) ext = a.end0;
      else {
        int lo = 0, hi = lc;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (pos_at(mid) > pos) hi = mid; else lo = mid + 1; }
        ext = pos_at(lo < lc ? lo : lc - 1);
      }
      const f32x4* e0 = reinterpret_cast<const f32x4*>(a.E0 + (long long)pos * a.D);
      const f32x4* e1 = reinterpret_cast<const f32x4*>(a.E1 + (long long)val * a.D);
      const f32x4* ex = reinterpret_cast<const f32x4*>(a.Ex + (long long)ext * a.D);
      const f32x4* pe = reinterpret_cast<const f32x4*>(a.pos_emb + (long long)(t - lc) * a.D);
      for (int qd = tid; qd < nq; qd += 256)
        *reinterpret_cast<f32x4*>(a.resid + pk_off(b, 4 * qd, a.D)) = ((e0[qd] + e1[qd]) + ex[qd]) + pe[qd];
    }
  }
}

// embedding of the token at position len[b]-1 (mingpt.py:256-286 + AR_N extra index) -> fragment-packed resid
__global__ __launch_bounds__(256) void embed_packed_kernel(const float* __restrict__ E0, const float* __restrict__ E1,
                                                           const float* __restrict__ Ex, const float* __restrict__ pos_emb,
                                                           const float* __restrict__ cond_pos_emb, const int* __restrict__ seq,
                                                           const int* __restrict__ len, const int* __restrict__ Lc,
                                                           float* __restrict__ resid, int D, int Lmax, int end0) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t = len[b] - 1, lc = Lc[b];
  const int* row = seq + (long long)b * Lmax * 2;
  const int pos = row[2 * t], val = row[2 * t + 1];
  int ext;
  if (t < lc) ext = pos;
  else if (pos == end0) ext = end0;
  else {
    int lo = 0, hi = lc;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (row[2 * mid] > pos) hi = mid; else lo = mid + 1; }
    ext = row[2 * (lo < lc ? lo : lc - 1)];
  }
  const f32x4* e0 = reinterpret_cast<const f32x4*>(E0 + (long long)pos * D);
  const f32x4* e1 = reinterpret_cast<const f32x4*>(E1 + (long long)val * D);
  const f32x4* ex = reinterpret_cast<const f32x4*>(Ex + (long long)ext * D);
  const f32x4* pe = reinterpret_cast<const f32x4*>(t < lc ? cond_pos_emb + (long long)t * D : pos_emb + (long long)(t - lc) * D);
  for (int qd = tid; qd < D / 4; qd += 256)
    *reinterpret_cast<f32x4*>(resid + pk_off(b, 4 * qd, D)) = ((e0[qd] + e1[qd]) + ex[qd]) + pe[qd];
}

// per-row cross entropy: loss[m] = logsumexp(logits[m,:V]) - logits[m, target[m]]   (F.cross_entropy, shapeformer.py:136)
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int* __restrict__ target,
                                                      float* __restrict__ loss, int V, int ld) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (long long)m * ld;
  float mx = -INFINITY;
  for (int v = tid; v < V; v += 256) mx = fmaxf(mx, row[v]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int v = tid; v < V; v += 256) se += __expf(row[v] - mx);
  se = wave_sum(se);
  if (lane == 0) red[4 + wave] = se;
  __syncthreads();
  if (tid == 0) loss[m] = mx + __logf((red[4] + red[5]) + (red[6] + red[7])) - row[target[m]];
}

// Decode-path weights from the raw parameters (load time, and after a training -> sampling switch): one workgroup per 16-row n-tile of
// W (N,K).  Wp = fragment order of W' = W diag(gamma) (gamma NULL: W itself); c1[n] = sum_k W'[n][k]; c2[n] = sum_k beta[k] W[n][k] + bias[n]
// (the constants of dgemm_kernel's LayerNorm form).  The two row sums are accumulated in float64 in a fixed order (16 partial sums of
// contiguous K ranges per row, added in range order) and rounded once: deterministic, and closer to the exact constants than an f32 sum.
__global__ __launch_bounds__(256) void ln_fold_pack_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ bias, float* __restrict__ Wp, float* __restrict__ c1,
                                                          float* __restrict__ c2, int N, int K) {
  __shared__ double p1[16][17], p2[16][17];
  const int nt = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 16 * K; i += 256) {      // fragment order [K/16][64 lanes][4]: lane = (k % 16 / 4) * 16 + row, j = k % 4
    const int k16 = i >> 8, l = (i >> 2) & 63, j = i & 3;
    const int n = nt * 16 + (l & 15), k = k16 * 16 + 4 * (l >> 4) + j;
    float w = n < N ? W[(long long)n * K + k] : 0.0f;
    if (gamma) w *= gamma[k];
    Wp[(long long)nt * 16 * K + i] = w;
  }
  if (!c1) return;
  const int r = tid >> 4, part = tid & 15, n = nt * 16 + r, kp = K / 16;
  double s1 = 0.0, s2 = 0.0;
  if (n < N)
    for (int k = part * kp; k < (part + 1) * kp; ++k) {
      const float w = W[(long long)n * K + k];
      s1 += (double)(gamma ? w * gamma[k] : w);      // the f32 product the GEMM multiplies with
      if (beta) s2 += (double)beta[k] * (double)w;
    }
  p1[r][part] = s1; p2[r][part] = s2;
  __syncthreads();
  if (part == 0) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < 16; ++q) { a += p1[r][q]; b += p2[r][q]; }
    if (n < N && bias) b += (double)bias[n];
    c1[nt * 16 + r] = (float)a;
    c2[nt * 16 + r] = (float)b;
  }
}

__global__ void set_len_kernel(int* len, const int* src, int B, int delta) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) len[i] = src[i] + delta;
}

// Host side of sfmi_decode_gemm_f32: picks the instantiation from the launch shape (and the explicit sfmi_tune_set knobs dgemm_nw /
// dgemm_un / dgemm_nt2 - nothing is read from the environment).  Only instantiations the default rules can reach exist: UN (k16-steps
// of loads in flight) is at most 8 / 4 / 2 / 2 / 2 / 1 for 1 .. 6 row tiles, which keeps every one of them free of scratch.
template <class P>
static int decode_gemm_launch(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid, float* out, int M,
                              int N, int K, int ldo, int ln, int act, int out_packed, int S, float* slab, int* cnt, int* pblk,
                              unsigned long long* prof, void* stream) {
  if (!x || !Wp16 || !out || M <= 0 || M > 192 || S <= 0 || K % S || (ln && !c1)) return SFMI_EINVAL;   // larger batches: several chains (gpt.py)
  if (out_packed && N % 16) return SFMI_EINVAL;
  if (S > 1 && (!slab || !cnt)) return SFMI_EINVAL;
  if (prof && !pblk) return SFMI_EINVAL;
  const int kslice = K / S;
  // up to 6 row tiles per workgroup; more rows = row groups (grid.z), each of MT tiles: the packed operands must hold groups * MT * 16 rows
  const int tiles = (M + 15) / 16, groups = (tiles + 5) / 6, MT = (tiles + groups - 1) / groups;
  int NWv = (kslice >= 2048 && MT <= 4) ? 16 : 8;
  const int knob_nw = g_tune.dgemm_nw;     // 0 = the rule above; 4 / 8 / 16 k-parts per workgroup where the shape allows it
  if ((knob_nw == 4 || knob_nw == 8 || knob_nw == 16) && kslice % (16 * knob_nw) == 0 && MT <= (knob_nw == 8 ? 6 : 4)) NWv = knob_nw;
  if (kslice % (16 * NWv)) NWv = kslice % 64 == 0 ? 4 : 1;   // narrow models (K-slice not a multiple of 128): fewer k-parts
  if (kslice % (16 * NWv)) return SFMI_EINVAL;
  DGemmArgs a;
  a.x = x; a.Wp = Wp16; a.c1 = c1; a.c2 = c2; a.resid = resid; a.out = out; a.M = M; a.N = N; a.K = K; a.ldo = ldo; a.ln = ln; a.act = act;
  a.out_packed = out_packed; a.slab = slab; a.cnt = cnt; a.pblk = pblk; a.prof = prof; a.prio = g_tune.dgemm_prio;
  hipStream_t st = (hipStream_t)stream;
  if ((g_tune.dgemm_nt2 == 2 || (g_tune.dgemm_nt2 == 1 && tiles % 3 == 0)) && NWv == 8 && (kslice / 8 / 16) % 2 == 0 && tiles >= 3) {
    // two n-tiles per wave, row groups of <= 3 row tiles, batches of two k16-steps: 5 operand loads per 24 MFMAs (7 in the one-tile
    // form) with the same 6 accumulator tiles per wave; same per-element arithmetic (k ascending, two chains): bit-identical.
    // Default (knob 1) when the row tiles divide into groups of exactly 3 (48 / 96 / 144 / 192 rows: no padded tile): GEMM phase of
    // 4 x 96 rows 3.06 -> 2.72 ms per step, loop 7.83 -> 7.53; with a padded tile (80 rows = 2 x 3 tiles for 5) it loses 1 %.
    const int g2 = (tiles + 2) / 3, MT2 = (tiles + g2 - 1) / g2;
    dim3 grid2(((N + 15) / 16 + 1) / 2, S, g2);
#define DG2(MT_) hipLaunchKernelGGL((dgemm_kernel<MT_, 8, 2, 2, P>), grid2, dim3(512), 0, st, a)
    if (MT2 == 1) DG2(1); else if (MT2 == 2) DG2(2); else DG2(3);
#undef DG2
    SFMI_CHECK_LAUNCH();
    return SFMI_OK;
  }
  dim3 grid((N + 15) / 16, S, groups);
  const int steps = kslice / NWv / 16;
  int un = MT == 1 ? 8 : (MT == 2 ? 4 : (MT <= 5 ? 2 : 1));   // UN weight + UN*MT activation float4 loads in flight per wave (5 row tiles: 128 VGPRs; 6: 138 with two)
  if (g_tune.dgemm_un > 0 && g_tune.dgemm_un < un) un = g_tune.dgemm_un;   // the knob can only lower it (deeper forms would spill)
  while (un > 1 && steps % un) un >>= 1;
#define DG(MT_, NW_, UN_) hipLaunchKernelGGL((dgemm_kernel<MT_, NW_, UN_, 1, P>), grid, dim3(64 * NW_), 0, st, a)
#define DGU8(MT_, NW_) do { if (un >= 8) DG(MT_, NW_, 8); else if (un >= 4) DG(MT_, NW_, 4); else if (un >= 2) DG(MT_, NW_, 2); else DG(MT_, NW_, 1); } while (0)
#define DGU4(MT_, NW_) do { if (un >= 4) DG(MT_, NW_, 4); else if (un >= 2) DG(MT_, NW_, 2); else DG(MT_, NW_, 1); } while (0)
#define DGU2(MT_, NW_) do { if (un >= 2) DG(MT_, NW_, 2); else DG(MT_, NW_, 1); } while (0)
  if (NWv == 16) { if (MT == 1) DGU8(1, 16); else if (MT == 2) DGU4(2, 16); else if (MT == 3) DGU2(3, 16); else DGU2(4, 16); }
  // (the epilogue operands are prefetched for row tile == wave, so a narrow launch needs MT <= NW)
  else if (NWv == 4) {
    if (MT == 1) DG(1, 4, 1); else if (MT == 2) DG(2, 4, 1);
    else if (MT == 3) DGU2(3, 4);
    else if (MT == 4) DGU2(4, 4);
    else return SFMI_EINVAL;
  }
  else if (NWv == 1) { if (MT == 1) DG(1, 1, 1); else return SFMI_EINVAL; }
  else if (MT == 1) DGU8(1, 8);
  else if (MT == 2) DGU4(2, 8);
  else if (MT <= 5) { if (MT == 3) DGU2(3, 8); else if (MT == 4) DGU2(4, 8); else DGU2(5, 8); }   // 65..96 rows: still the 8-wave kernel
  else DG(6, 8, 1);
#undef DGU8
#undef DGU4
#undef DGU2
#undef DG
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

extern "C" {

// host: Linear weight (N,K) row-major -> 16x16x4-MFMA fragment order Wp16 = [ceil(N/16)][K/16][64][4] (rows >= N zero)
size_t sfmi_skinny16_pack_floats(int N, int K) { return (size_t)((N + 15) / 16) * 16 * K; }
int sfmi_skinny16_pack_weight(const float* W, int N, int K, float* out) {
  if (!W || !out || K % 16) return SFMI_EINVAL;
  const int NT = (N + 15) / 16;
  for (int nt = 0; nt < NT; ++nt)
    for (int k16 = 0; k16 < K / 16; ++k16)
      for (int l = 0; l < 64; ++l) {
        const int n = nt * 16 + (l & 15);
        for (int j = 0; j < 4; ++j) {
          const int k = k16 * 16 + 4 * (l >> 4) + j;
          out[(((size_t)nt * (K / 16) + k16) * 64 + l) * 4 + j] = n < N ? W[(size_t)n * K + k] : 0.0f;
        }
      }
  return SFMI_OK;
}
// device form of sfmi_skinny16_pack_weight with the LayerNorm fold of dgemm_kernel's header: W (N,K) row-major, gamma / beta (K) of the
// LayerNorm in front of the Linear (mingpt.py:103-111; both NULL: plain pack), bias (N) or NULL -> Wp (ceil(N/16)*16*K floats), c1 / c2
// (ceil(N/16)*16 floats each; NULL with a plain pack).  No library GEMV / reduction on the load path.
int sfmi_ln_fold_pack_f32(const float* W, const float* gamma, const float* beta, const float* bias, float* Wp, float* c1, float* c2, int N,
                          int K, void* stream) {
  if (!W || !Wp || N <= 0 || K <= 0 || K % 16 || (!c1) != (!c2) || ((gamma || beta) && !c1)) return SFMI_EINVAL;
  hipLaunchKernelGGL(ln_fold_pack_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, W, gamma, beta, bias, Wp, c1, c2, N, K);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// replaces LayerNorm + nn.Linear (+GELU / +residual) of Block.forward at decode time (mingpt.py:103-111).
// Wp16: sfmi_skinny16_pack_weight of W (plain) or of W*diag(gamma) (ln=1, with c1/c2 as in the kernel header).
// c1/c2 need ceil(N/16)*16 readable floats.  resid (if given) is added and shares out's (M,ldo) layout.
// x (and out/resid when out_packed) are fragment-packed [ceil(M/16)][N/16][64][4] (see pk_off); out_packed == 0
// writes row-major (M,ldo) (used for the logits handed to the sampler).  S > 1 splits K across S workgroups per
// n-tile with an in-kernel deterministic last-arriver reduction (slab/cnt scratch, cnt zero-initialised ONCE).
size_t sfmi_decode_gemm_slab_floats(int M, int N, int S) { return (size_t)((M + 63) / 64 * 4 + 2) * ((N + 15) / 16) * S * 320; }
// rows the fragment-packed operands of sfmi_decode_gemm_f32 must hold for M rows: groups * MT * 16 (M <= 96: ceil(M/16)*16)
int sfmi_decode_gemm_padded_rows(int M) {
  const int tiles = (M + 15) / 16, groups = (tiles + 5) / 6, MT = (tiles + groups - 1) / groups;
  const int g2 = (tiles + 2) / 3, MT2 = (tiles + g2 - 1) / g2;       // the two-n-tile form (dgemm_nt2): groups of <= 3 row tiles
  return max(groups * MT, g2 * MT2) * 16;
}
int sfmi_decode_gemm_prof_f32(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid,
                              float* out, int M, int N, int K, int ldo, int ln, int act, int out_packed, int S, float* slab,
                              int* cnt, int* pblk, unsigned long long* prof, void* stream) {
  return decode_gemm_launch<DgProduct>(x, Wp16, c1, c2, resid, out, M, N, K, ldo, ln, act, out_packed, S, slab, cnt, pblk, prof, stream);
}
int sfmi_decode_gemm_f32(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid,
                         float* out, int M, int N, int K, int ldo, int ln, int act, int out_packed, int S, float* slab,
                         int* cnt, void* stream) {
  return decode_gemm_launch<DgProduct>(x, Wp16, c1, c2, resid, out, M, N, K, ldo, ln, act, out_packed, S, slab, cnt, nullptr, nullptr, stream);
}

// replaces get_embeddings (mingpt.py:256-286) + the AR_N extra index (representers.py:188-196,432-442)
// (+ LayerNorm ln1 of the first block).  P == 0: one row per sequence at t = len[b]-1; P > 0: prefill rows (b,t<P).
int sfmi_gpt_embed_f32(const float* E0, const float* E1, const float* Ex, const float* pos_emb, const float* cond_pos_emb,
                       const int* seq, const int* len, const int* Lc, const int* nval, const int* extra, int* extra_out,
                       float* resid_out, float* xn, const float* gamma, const float* beta, int B, int P, int D, int Lmax,
                       int end0, const int* rowoff, int M_packed, void* stream) {
  if (!E0 || !E1 || !Ex || !pos_emb || !cond_pos_emb || !seq || !len || !Lc || D % 4 || D > 4096) return SFMI_EINVAL;
  if (rowoff && (P <= 0 || M_packed <= 0)) return SFMI_EINVAL;
  RowPrepArgs a = {};
  a.E0 = E0; a.E1 = E1; a.Ex = Ex; a.pos_emb = pos_emb; a.cond_pos_emb = cond_pos_emb; a.seq = seq; a.len = len; a.Lc = Lc;
  a.nval = nval; a.extra = extra; a.extra_out = extra_out;
  a.resid_out = resid_out; a.xn = xn; a.gamma = gamma; a.beta = beta; a.mode = 0; a.M = P ? B * P : B; a.D = D; a.Lmax = Lmax;
  a.P = P; a.end0 = end0; a.rowoff = rowoff; a.nB = B;
  if (rowoff) a.M = M_packed;
  hipLaunchKernelGGL(rowprep_kernel, dim3(a.M), dim3(256), 0, (hipStream_t)stream, a);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces the residual adds + LayerNorm of Block.forward (mingpt.py:107-111): x = resid + sum_s part[s] + bias
// (+ tok_embs[0][next pos], mingpt.py:294) ; resid_out = x ; xn = LN(x)
int sfmi_gpt_rowprep_f32(const float* resid_in, const float* part, const float* bias, const float* Eadd, const int* seq,
                         const int* len, const int* Lc, const int* nval, float* resid_out, float* xn, const float* gamma,
                         const float* beta, int S, int M, int P, int D, int Lmax, const int* rowoff, int B, void* stream) {
  if (!resid_in || D % 4 || D > 4096 || (Eadd && (!seq || !len)) || (rowoff && B <= 0)) return SFMI_EINVAL;
  RowPrepArgs a = {};
  a.resid_in = resid_in; a.part = part; a.bias = bias; a.Eadd = Eadd; a.seq = seq; a.len = len; a.Lc = Lc; a.nval = nval;
  a.resid_out = resid_out; a.xn = xn; a.gamma = gamma; a.beta = beta; a.mode = 1; a.S = S; a.M = M; a.D = D; a.Lmax = Lmax; a.P = P;
  a.rowoff = rowoff; a.nB = B;
  hipLaunchKernelGGL(rowprep_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, a);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces CausalSelfAttention.forward for ONE new position per row with a KV cache (mingpt.py:73-91).
// sem (optional, 3 device ints zeroed by the caller while no launch is in flight) + blk (1 device int, zero; one per chain):
// the launch passes the attention turnstile first (at most `lanes` gated launches stream at a time, FIFO).
int sfmi_gpt_attn_decode_gated_f32(const float* qkv_part, float* Kc, float* Vc, const int* len, float* y, int B, int D, int H,
                                   int Lmax, const int* shared_len, int* sem, int* blk, int lanes, unsigned long long* prof, void* stream) {
  if (!qkv_part || !Kc || !Vc || !len || !y || D % H || (D / H) > 64 || (D / H) % 4 || Lmax > 1024) return SFMI_EINVAL;
  if (sem && (!blk || lanes <= 0)) return SFMI_EINVAL;
  if (prof && !blk) return SFMI_EINVAL;
  AttnArgs a;
  a.qkv = qkv_part; a.Kc = Kc; a.Vc = Vc; a.len = len; a.y = y; a.shared_len = shared_len;
  a.B = B; a.H = H; a.D = D; a.Lmax = Lmax; a.HD = D / H; a.scale = 1.0f / sqrtf((float)a.HD);
  a.sem = sem; a.blk = blk; a.prof = prof;
  const int nitems = B * H;
  const int grid = g_tune.attn_blocks > 0 ? min(g_tune.attn_blocks, nitems) : nitems;
  const size_t pad = (size_t)g_tune.attn_lds_pad;
  hipStream_t st = (hipStream_t)stream;
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [] {   // the occupancy-cap experiments ask for more dynamic LDS than the 64 KB default
#define AT_ATTR(W_, U_) do { hipError_t e_ = hipFuncSetAttribute((const void*)attn_decode_kernel<W_, U_, W_ != 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); if (e_ != hipSuccess) attr_err = e_; \
                            e_ = hipFuncSetAttribute((const void*)attn_decode_kernel<W_, U_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024); if (e_ != hipSuccess) attr_err = e_; } while (0)
    AT_ATTR(16, 2); AT_ATTR(16, 4); AT_ATTR(16, 8); AT_ATTR(8, 2); AT_ATTR(8, 4); AT_ATTR(8, 8); AT_ATTR(4, 8); AT_ATTR(4, 16);
#undef AT_ATTR
  });
  if (pad && attr_err != hipSuccess) return (int)attr_err;
  if (sem) hipLaunchKernelGGL(attn_gate_kernel, dim3(1), dim3(64), 0, st, sem, lanes);
  // the plain instance exists for the 16-wave shapes (the product's); the 8- / 4-wave experiment shapes keep the general one (the
  // 8-wave plain instance spilled 12 bytes)
#define AT(W_, U_) do { if (a.shared_len || W_ != 16) hipLaunchKernelGGL((attn_decode_kernel<W_, U_, true>), dim3(grid), dim3(64 * W_), pad, st, a); \
                        else hipLaunchKernelGGL((attn_decode_kernel<W_, U_, W_ != 16>), dim3(grid), dim3(64 * W_), pad, st, a); } while (0)
  if (g_tune.attn_waves == 16) { if (g_tune.attn_unroll == 8) AT(16, 8); else if (g_tune.attn_unroll == 2) AT(16, 2); else AT(16, 4); }
  else if (g_tune.attn_waves == 4) { if (g_tune.attn_unroll == 16) AT(4, 16); else AT(4, 8); }      // light-occupancy experiment (round 5)
  else { if (g_tune.attn_unroll == 8) AT(8, 8); else if (g_tune.attn_unroll == 2) AT(8, 2); else AT(8, 4); }
#undef AT
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_gpt_attn_decode_f32(const float* qkv_part, const float* bqkv, float* Kc, float* Vc, const int* len, float* y,
                             int S, int B, int D, int H, int Lmax, const int* shared_len, void* stream) {
  if (!bqkv) return SFMI_EINVAL;
  return sfmi_gpt_attn_decode_gated_f32(qkv_part, Kc, Vc, len, y, B, D, H, Lmax, shared_len, nullptr, nullptr, 0, nullptr, stream);
}

// causal self-attention over the conditioning prefix (positions 0..Lc[b]-2), also fills the KV caches
int sfmi_gpt_attn_prefill_lse_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                                  int Lmax, const int* rowoff, float drop_p, unsigned drop_seed, float* lse, void* stream);
int sfmi_gpt_attn_prefill_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                              int Lmax, const int* rowoff, float drop_p, unsigned drop_seed, void* stream) {
  return sfmi_gpt_attn_prefill_lse_f32(qkv, Kc, Vc, nval, y, B, P, D, H, Lmax, rowoff, drop_p, drop_seed, nullptr, stream);
}
// the same launch; lse != NULL also receives the (B,H,P) row log-sum-exps of the scaled scores (what sfmi_attn_bwd_lse_f32 starts from)
int sfmi_gpt_attn_prefill_lse_sd_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                                     int Lmax, const int* rowoff, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, float* lse,
                                     void* stream);
int sfmi_gpt_attn_prefill_lse_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                                  int Lmax, const int* rowoff, float drop_p, unsigned drop_seed, float* lse, void* stream) {
  return sfmi_gpt_attn_prefill_lse_sd_f32(qkv, Kc, Vc, nval, y, B, P, D, H, Lmax, rowoff, drop_p, drop_seed, nullptr, lse, stream);
}
// the same with the dropout seed optionally in device memory (drop_seed_dev != NULL overrides drop_seed at run time)
int sfmi_gpt_attn_prefill_lse_sd_f32(const float* qkv, float* Kc, float* Vc, const int* nval, float* y, int B, int P, int D, int H,
                                     int Lmax, const int* rowoff, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, float* lse,
                                     void* stream) {
  const int HD = H > 0 ? D / H : 0;
  if (!qkv || !Kc || !Vc || !nval || !y || H <= 0 || D % H || (HD != 16 && HD != 32 && HD != 64) || P <= 0 || drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  const dim3 grid(B, H, (P + 63) / 64);
  const float scale = 1.0f / sqrtf((float)HD);
  hipStream_t st = (hipStream_t)stream;
  if (HD == 64) hipLaunchKernelGGL(attn_prefill_mfma_kernel<64>, grid, dim3(256), 0, st, qkv, Kc, Vc, nval, y, P, D, Lmax, scale, rowoff, drop_p, drop_seed, lse, drop_seed_dev);
  else if (HD == 32) hipLaunchKernelGGL(attn_prefill_mfma_kernel<32>, grid, dim3(256), 0, st, qkv, Kc, Vc, nval, y, P, D, Lmax, scale, rowoff, drop_p, drop_seed, lse, drop_seed_dev);
  else hipLaunchKernelGGL(attn_prefill_mfma_kernel<16>, grid, dim3(256), 0, st, qkv, Kc, Vc, nval, y, P, D, Lmax, scale, rowoff, drop_p, drop_seed, lse, drop_seed_dev);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces sampling_masker + sample_logits for one tuple element (representers.py:120-155, common.py:260-299,
// shapeformer.py:91-106); advance != 0 also appends the token (len += 1).  Optional fused tail (resid != NULL):
// tuple 0: resid[b] += E0[pos'] ; tuple 1: resid[b] = embedding of the completed token (next step's input).
int sfmi_gpt_sample_f32(const float* part, int* seq, int* len, const int* Lc, float* logp, float* hist, const int* force,
                        float* resid, const float* E0, const float* E1, const float* Ex, const float* pos_emb, int D,
                        int S, int B, int V, int ldv, int Lmax, int tuple_i, int end0, int end1, int top_k, float top_p,
                        float temperature, int greedy_row0, int mask_invalid, int mask_completion, int max_steps,
                        unsigned seed, const unsigned* seed_dev, int advance, int row_offset, int rows_total, int step_offset,
                        void* stream) {
  if (!part || !seq || !len || !Lc || V > 4352 || temperature <= 0.f || rows_total < B + row_offset || step_offset < 0) return SFMI_EINVAL;
  if (resid && (!E0 || (tuple_i == 1 && (!E1 || !Ex || !pos_emb)) || D % 4)) return SFMI_EINVAL;
  SampleArgs a;
  a.part = part; a.seq = seq; a.len = len; a.Lc = Lc; a.logp = logp; a.hist = hist; a.force = force; a.S = S; a.M = B; a.V = V;
  a.ldv = ldv; a.resid = resid; a.E0 = E0; a.E1 = E1; a.Ex = Ex; a.pos_emb = pos_emb; a.D = D;
  a.Lmax = Lmax; a.tuple_i = tuple_i; a.end0 = end0; a.end1 = end1; a.top_k = top_k; a.greedy_row0 = greedy_row0;
  a.mask_invalid = mask_invalid; a.mask_completion = mask_completion; a.max_steps = max_steps; a.advance = advance;
  a.top_p = top_p; a.temperature = temperature; a.seed = seed; a.seed_dev = seed_dev; a.row_offset = row_offset; a.rows_total = rows_total;
  a.mask_out = nullptr; a.step_offset = step_offset;
  const size_t dyn = (top_k <= 0 || top_k > SMP_MAXC) ? (size_t)SMP_BIG * 12 : 0;
  hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(256), dyn, (hipStream_t)stream, a);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces ShapeRepresenter.sampling_masker alone (representers.py:120-155): logits (B,ldv) -> masked copy out (B,V); the masking
// stage of sample_kernel and nothing else (no draw, seq / len untouched).  seq (B,Lmax,2): row b holds len[b] complete tokens and,
// for tuple_i == 1, the position just drawn at seq[b][len[b]][0].
int sfmi_gpt_mask_logits_f32(const float* logits, const int* seq, const int* len, const int* Lc, float* out, int B, int V, int ldv,
                             int Lmax, int tuple_i, int end0, int end1, int mask_invalid, int mask_completion, void* stream) {
  if (!logits || !seq || !len || !Lc || !out || B <= 0 || V <= 0 || V > 4352 || ldv < V || (tuple_i != 0 && tuple_i != 1)) return SFMI_EINVAL;
  SampleArgs a = {};
  a.part = logits; a.seq = const_cast<int*>(seq); a.len = const_cast<int*>(len); a.Lc = Lc; a.S = 1; a.M = B; a.V = V; a.ldv = ldv;
  a.Lmax = Lmax; a.tuple_i = tuple_i; a.end0 = end0; a.end1 = end1; a.top_k = 1; a.mask_invalid = mask_invalid;
  a.mask_completion = mask_completion; a.max_steps = 1; a.temperature = 1.0f; a.rows_total = B; a.mask_out = out;
  hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// decode-path embedding into the fragment-packed residual buffer (input of the first decode step)
int sfmi_gpt_embed_packed_f32(const float* E0, const float* E1, const float* Ex, const float* pos_emb, const float* cond_pos_emb,
                              const int* seq, const int* len, const int* Lc, float* resid, int B, int D, int Lmax, int end0,
                              void* stream) {
  if (!E0 || !E1 || !Ex || !pos_emb || !cond_pos_emb || !seq || !len || !Lc || !resid || D % 16) return SFMI_EINVAL;
  hipLaunchKernelGGL(embed_packed_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, E0, E1, Ex, pos_emb, cond_pos_emb, seq, len, Lc,
                     resid, D, Lmax, end0);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces F.cross_entropy(reduction='none') over rows (shapeformer.py:132-140)
int sfmi_ce_rows_f32(const float* logits, const int* target, float* loss, long long M, int V, int ld, void* stream) {
  if (!logits || !target || !loss || M <= 0 || V <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, logits, target, loss, V, ld);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_set_len_i32(int* len, const int* src, int B, int delta, void* stream) {
  if (!len || !src) return SFMI_EINVAL;
  hipLaunchKernelGGL(set_len_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, len, src, B, delta);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
