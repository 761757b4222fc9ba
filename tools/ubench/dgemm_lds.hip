// EXPERIMENT (round 2, not in libsfmi): LDS-staged decode GEMM for 17..96 rows (gfx950, f32 MFMA 16x16x4) as a candidate
// replacement of dgemm_kernel (csrc/gpt.hip) on the
// KV-cached decode step of CondTupleGPT (LayerNorm + nn.Linear (+GELU / +residual) of Block.forward, mingpt.py:103-111).
//
// Why: dgemm_kernel gives every workgroup ONE 16-column n-tile over the full K, so each of the N/16 workgroups pulls the
// whole (M x K) activation operand through its CU's L1 (fc1 at 64 rows: 256 workgroups x 256 KB = 64 MB of L2->L1 fills
// for a 16 MB weight matrix) and the kernel is bound by that fill rate (~25 B/clk/CU), not by MFMA or HBM
// (profiles/r01_pmc_sq_dgemm_M64.txt: MFMA busy 14 %).  Here a workgroup owns NB adjacent n-tiles x ONE K-slice:
//
//   * the (M x K/S) activation slice is copied ONCE per workgroup into LDS (fragment-packed pieces are 1 KiB
//     contiguous, the copy is verbatim, lane-linear, conflict-free) and read from there by all NB n-tile waves
//     (ds_read_b128): L2->CU activation traffic drops by NB x, per-CU need is 32/NB B/clk at full MFMA rate;
//   * wave (nb, kp) owns n-tile nb x k-part kp (SW k16-steps): its weights are SW float4 registers, ALL requested up
//     front (the weight stream of the whole launch is in flight from t = 0; MFMAs start when the first step lands);
//   * K-parts are summed through LDS in fixed order; K-slices (S > 1) through write-through slabs + a per-tile ticket,
//     last arriver sums in slice order (deterministic; same slab / ticket protocol as dgemm_kernel, guide G16).
//   grid (ceil(ceil(N/16)/NB), S), NB*KP waves.  K/S = 16*KP*SW.
#include "sfmi_common.h"

struct DGemmLdsArgs {
  const float* x; const float* Wp; const float* c1; const float* c2; const float* resid; float* out;
  int M, N, K, ldo, ln, act, out_packed;
  float* slab; int* cnt;
};

__device__ __forceinline__ void dl_st_sc1(float* p, f32x4 v) {
  unsigned long long lo = ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]);
  unsigned long long hi = ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 dl_ld_sc1(const float* p) {
  unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f32x4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
               __uint_as_float((unsigned)(hi >> 32))};
}

template <int MT, int NB, int KP, int SW>
__global__ __launch_bounds__(64 * NB * KP, (NB * KP >= 16 ? 4 : (NB * KP > 8 ? 3 : (SW >= 8 ? 2 : 4)))) void dgemm_lds_kernel(DGemmLdsArgs a) {
  constexpr int NW = NB * KP;            // waves
  constexpr int ST = KP * SW;            // k16-steps of this workgroup's K-slice
  constexpr int XP = (MT * SW + NB - 1) / NB;   // activation pieces staged per wave (its own k-part, dealt over the NB waves)
  constexpr int TPW = (NB * MT + NW - 1) / NW;  // output tiles finished per wave
  extern __shared__ __attribute__((aligned(16))) float dl_lds[];
  f32x4* xs = reinterpret_cast<f32x4*>(dl_lds);                 // [MT][ST][64] activation slice (phase 1)
  f32x4* red = reinterpret_cast<f32x4*>(dl_lds);                // [KP][NB][MT][64] partial tiles (phase 2, aliases xs)
  float* stat = dl_lds + (size_t)KP * NB * MT * 256;            // [KP][2][MT][16] LayerNorm partial sums
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, ml = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: addresses become SGPR base + lane offset
  const int nb = wave % NB, kp = wave / NB;
  const int cb = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
  const int ntiles = (a.N + 15) >> 4;
  const int kt = a.K >> 4;
  const int s0 = sp * ST + kp * SW;      // first k16-step of this wave
  const int nt = min(cb * NB + nb, ntiles - 1);
  // ---- all loads of the wave are requested up front: activation pieces (L2), then the weights (HBM) -----------------
  f32x4 xr[XP];
  const f32x4* x4 = reinterpret_cast<const f32x4*>(a.x) + lane;
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = min(nb + i * NB, MT * SW - 1), j = p / SW, s = p % SW;
#ifdef DL_ABL_NOX
    xr[i] = f32x4{(float)lane, 1.f, 2.f, (float)i};
#else
    xr[i] = x4[((long long)j * kt + s0 + s) * 64];
#endif
  }
  f32x4 w[SW];
  const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp) + ((long long)nt * kt + s0) * 64 + lane;
#ifdef DL_ABL_NOWLOAD   // ablation (tools/ubench): no HBM weight stream
#pragma unroll
  for (int s = 0; s < SW; ++s) w[s] = f32x4{(float)lane, 1.f, 2.f, (float)s};
#else
#pragma unroll
  for (int s = 0; s < SW; ++s) w[s] = wp[s * 64];
#endif
  __builtin_amdgcn_sched_barrier(0);
  // ---- stage the activation slice ---------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int p = min(nb + i * NB, MT * SW - 1), j = p / SW, s = p % SW;
    xs[(j * ST + kp * SW + s) * 64 + lane] = xr[i];
  }
  __syncthreads();
  // epilogue operands of the tiles this wave finishes (tile t = wave + i*NW -> n-tile t % NB, row tile t / NB): requested
  // here (the staging registers are free again, the weights are still in flight), UNCONDITIONALLY from clamped addresses (absent operands read a valid dummy) - a branch around a load makes
  // hipcc drain vmcnt(0) at the join, i.e. wait for the whole weight stream before the first MFMA
  f32x4 pc1[TPW], pc2[TPW], pres[TPW];
  {
    const float* c1p = a.ln ? a.c1 : a.Wp;
    const float* c2p = a.c2 ? a.c2 : a.Wp;
    const float* rp = a.resid ? a.resid : a.out;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int t = min(wave + i * NW, NB * MT - 1);
      const int tnt = min(cb * NB + t % NB, ntiles - 1), j = t / NB;
      int n_ep = tnt * 16 + 4 * q;
      pc1[i] = *reinterpret_cast<const f32x4*>(c1p + n_ep);
      pc2[i] = *reinterpret_cast<const f32x4*>(c2p + n_ep);
      if (!a.out_packed && n_ep + 4 > a.ldo) n_ep = 0;
      const long long off = a.out_packed ? (((long long)j * (a.N >> 4) + tnt) * 64 + lane) * 4
                                         : (long long)min(j * 16 + ml, a.M - 1) * a.ldo + n_ep;
      pres[i] = *reinterpret_cast<const f32x4*>(rp + off);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- MFMA: SW k16-steps x MT row tiles, B operand from LDS ------------------------------------------------------
  f32x4 acc[MT];
  float s1[MT], s2[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) { acc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[j] = 0.f; s2[j] = 0.f; }
  const f32x4* xw = xs + (kp * SW) * 64 + lane;
  // B fragments one k16-step ahead of their MFMAs (two register sets); the step boundaries are scheduling barriers so
  // that hipcc neither hoists every ds_read to the top (128 extra VGPRs) nor sinks them next to their uses
  f32x4 xv[2][MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) xv[0][j] = xw[(j * ST) * 64];
#pragma unroll
  for (int s = 0; s < SW; ++s) {
    if (s + 1 < SW) {
#pragma unroll
      for (int j = 0; j < MT; ++j) xv[(s + 1) & 1][j] = xw[(j * ST + s + 1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const f32x4 v = xv[s & 1][j];
      s1[j] += (v[0] + v[1]) + (v[2] + v[3]);
      s2[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
#ifdef DL_ABL_NOMFMA
      for (int j = 0; j < MT; ++j) acc[j][e] += w[s][e] * xv[s & 1][j][e];
#else
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s][e], xv[s & 1][j][e], acc[j], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();     // every wave is done with xs: the region becomes the reduction buffer
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    red[((kp * NB + nb) * MT + j) * 64 + lane] = acc[j];
    if (a.ln && nb == 0) {
      float t1 = s1[j], t2 = s2[j];
      t1 += __shfl_xor(t1, 16, 64); t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 16, 64); t2 += __shfl_xor(t2, 32, 64);
      if (q == 0) { stat[((kp * 2 + 0) * MT + j) * 16 + ml] = t1; stat[((kp * 2 + 1) * MT + j) * 16 + ml] = t2; }
    }
  }
  __syncthreads();
  // ---- finish tiles: sum the k-parts in order, split-K hand-off, epilogue -----------------------------------------
  f32x4 r[TPW];
  float t1[TPW], t2[TPW];
  bool live[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wave + i * NW;
    live[i] = t < NB * MT && cb * NB + t % NB < ntiles;
    r[i] = f32x4{0.f, 0.f, 0.f, 0.f}; t1[i] = 0.f; t2[i] = 0.f;
    if (t < NB * MT) {
      const int tn = t % NB, j = t / NB;
#pragma unroll
      for (int k = 0; k < KP; ++k) r[i] = r[i] + red[((k * NB + tn) * MT + j) * 64 + lane];
      if (a.ln) {
#pragma unroll
        for (int k = 0; k < KP; ++k) { t1[i] += stat[((k * 2 + 0) * MT + j) * 16 + ml]; t2[i] += stat[((k * 2 + 1) * MT + j) * 16 + ml]; }
      }
    }
  }
#ifdef DL_ABL_NOTAIL
  if (false) {
#else
  if (S > 1) {
#endif
    // publish this slice's tiles write-through, one ticket per tile (all of the wave's tickets in ONE atomic round trip);
    // the last arriver of a tile sums its S slabs in slice order and runs the epilogue
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if (!live[i]) continue;
      const int t = wave + i * NW;
      const long long tile = (long long)(t / NB) * ntiles + cb * NB + t % NB;
      float* slab = a.slab + (tile * S + sp) * 320;          // 256 acc floats + 2 x 16 stat floats
      dl_st_sc1(slab + lane * 4, r[i]);
      if (a.ln && q == 0) {
        __hip_atomic_store(slab + 256 + ml, t1[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slab + 272 + ml, t2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int ticket = 0;
    {
      const int t = wave + lane * NW;                         // lane i takes the ticket of the wave's i-th tile
      if (lane < TPW && t < NB * MT && cb * NB + t % NB < ntiles) {
        const long long tile = (long long)(t / NB) * ntiles + cb * NB + t % NB;
        ticket = __hip_atomic_fetch_add(a.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == S - 1) __hip_atomic_store(a.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int tk = __shfl(ticket, i, 64);
      live[i] = live[i] && tk == S - 1;
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      if (!live[i]) continue;
      const int t = wave + i * NW;
      const long long tile = (long long)(t / NB) * ntiles + cb * NB + t % NB;
      const float* base = a.slab + tile * S * 320;
      r[i] = f32x4{0.f, 0.f, 0.f, 0.f}; t1[i] = 0.f; t2[i] = 0.f;
      for (int s = 0; s < S; s += 4) {   // four slices in flight per round trip, summed in slice order
        f32x4 tv[4];
        float u1[4], u2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* b = base + min(s + u, S - 1) * 320;
          tv[u] = dl_ld_sc1(b + lane * 4);
          u1[u] = __hip_atomic_load(b + 256 + ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          u2[u] = __hip_atomic_load(b + 272 + ml, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (s + u < S) { r[i] = r[i] + tv[u]; t1[i] += u1[u]; t2[i] += u2[u]; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (!live[i]) continue;
    const int t = wave + i * NW;
    const int tnt = cb * NB + t % NB, j = t / NB;
    const int m = j * 16 + ml, n = tnt * 16 + 4 * q;
    if (!((a.out_packed || m < a.M) && n < a.N)) continue;
    f32x4 v = r[i];
    if (a.ln) {
      const float mean = t1[i] / (float)a.K;
      const float var = fmaxf(t2[i] / (float)a.K - mean * mean, 0.f);
      const float rstd = rsqrtf(var + 1e-5f);
      v = (v - pc1[i] * mean) * rstd;
    }
    if (a.c2) v = v + pc2[i];
    if (a.act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
    }
    const long long off = a.out_packed ? (((long long)j * (a.N >> 4) + tnt) * 64 + lane) * 4 : (long long)m * a.ldo + n;
    if (a.resid) v = v + pres[i];
    *reinterpret_cast<f32x4*>(a.out + off) = v;
  }
}

template <int MT, int NB, int KP, int SW>
static int dl_launch(const DGemmLdsArgs& a, int S, hipStream_t st) {
  constexpr int NW = NB * KP, ST = KP * SW;
  constexpr size_t xs_b = (size_t)MT * ST * 1024, red_b = (size_t)KP * NB * MT * 1024 + (size_t)KP * 2 * MT * 64;
  constexpr size_t lds = xs_b > red_b ? xs_b : red_b;
  if constexpr (lds <= 160 * 1024 && NW <= 16) {
    static bool attr_set = false;   // idempotent, race-free
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)dgemm_lds_kernel<MT, NB, KP, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    const int ntiles = (a.N + 15) / 16;
    hipLaunchKernelGGL((dgemm_lds_kernel<MT, NB, KP, SW>), dim3((ntiles + NB - 1) / NB, S), dim3(64 * NW), lds, st, a);
    return SFMI_OK;
  } else {
    return SFMI_EINVAL;   // activation slice does not fit the 160 KiB LDS
  }
}

extern "C" {

// Same contract as sfmi_decode_gemm_f32 (csrc/gpt.hip; operands fragment-packed, LayerNorm folded, deterministic
// split-K) with the workgroup shape chosen by the caller: NB n-tiles x (K/S) slice per workgroup, KP k-parts.
// Supported: M <= 96 rows; (NB, KP) in {(4,2), (2,4), (3,4), (4,4), (2,8), (1,8)}; K/S/16/KP in {2, 4, 8}.
int sfmi_decode_gemm_lds_f32(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid,
                             float* out, int M, int N, int K, int ldo, int ln, int act, int out_packed, int S, int NB,
                             int KP, float* slab, int* cnt, void* stream) {
  if (!x || !Wp16 || !out || M <= 0 || M > 96 || S <= 0 || K % (S * 16 * KP) || (ln && !c1)) return SFMI_EINVAL;
  if (out_packed && N % 16) return SFMI_EINVAL;
  if (S > 1 && (!slab || !cnt)) return SFMI_EINVAL;
  const int SW = K / S / 16 / KP, MT = (M + 15) / 16;
  DGemmLdsArgs a;
  a.x = x; a.Wp = Wp16; a.c1 = c1; a.c2 = c2; a.resid = resid; a.out = out; a.M = M; a.N = N; a.K = K; a.ldo = ldo; a.ln = ln;
  a.act = act; a.out_packed = out_packed; a.slab = slab; a.cnt = cnt;
  hipStream_t st = (hipStream_t)stream;
#define DL_SW(MT_, NB_, KP_)                                                         \
  do {                                                                               \
    int rc_ = SFMI_EINVAL;                                                           \
    if (SW == 2) rc_ = dl_launch<MT_, NB_, KP_, 2>(a, S, st);                        \
    else if (SW == 4) rc_ = dl_launch<MT_, NB_, KP_, 4>(a, S, st);                   \
    else if (SW == 8) rc_ = dl_launch<MT_, NB_, KP_, 8>(a, S, st);                   \
    if (rc_ != SFMI_OK) return rc_;                                                  \
  } while (0)
#define DL_MT(NB_, KP_)                                                              \
  do {                                                                               \
    switch (MT) {                                                                    \
      case 1: DL_SW(1, NB_, KP_); break;                                             \
      case 2: DL_SW(2, NB_, KP_); break;                                             \
      case 3: DL_SW(3, NB_, KP_); break;                                             \
      case 4: DL_SW(4, NB_, KP_); break;                                             \
      case 5: DL_SW(5, NB_, KP_); break;                                             \
      default: DL_SW(6, NB_, KP_); break;                                            \
    }                                                                                \
  } while (0)
  if (NB == 4 && KP == 2) DL_MT(4, 2);
  else if (NB == 2 && KP == 4) DL_MT(2, 4);
  else if (NB == 3 && KP == 4) DL_MT(3, 4);
  else if (NB == 4 && KP == 4) DL_MT(4, 4);
  else if (NB == 2 && KP == 8) DL_MT(2, 8);
  else if (NB == 1 && KP == 8) DL_MT(1, 8);
  else return SFMI_EINVAL;
#undef DL_MT
#undef DL_SW
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
