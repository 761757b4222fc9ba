// Work-balanced ("stream-K") f32 GEMM on the matrix cores for products too SMALL to fill 256 CUs with whole output tiles: the Linear
// layers of the transformer's TRAINING step at the YAML's batch size (mingpt.py:46-111 and their autograd at M = B*L ~ 500 rows:
// y = x W^T, dX = dY W, dW = dY^T X).
//
//   C (M,N; ldc) = op(A) (M,K) * op(B) (K,N)  [+ C]  [+ bias[n]] [act] [dropout] [+ resid]
//
// csrc/sgemm.hip gives every 128 x 128 output tile to one workgroup (plus a power-of-two split of K and a second launch that adds
// the slices).  At 500 rows that leaves 32-128 tiles for 512 workgroup slots: 25-75 % of the CUs idle, and ~290 reduction launches
// per step.  Here the unit of work is ONE K-CHUNK OF ONE TILE; the T * C units are dealt out evenly, in order, to a grid of
// G = 2 x 256 workgroups, so every CU executes the same number of MFMAs whatever the shape.  A workgroup walks its unit range tile by
// tile ("segments").  A tile whose chunks all fall to one workgroup takes the direct epilogue.  A tile cut between workgroups is
// finished without a second launch and without spinning: each WAVE publishes the accumulators of its own sub-tile write-through
// (16-byte `sc1` stores into the workgroup's slot of the slab, lane-linear = the register layout), drains them, takes a ticket on the
// (tile, wave) counter; the wave that draws the last ticket re-arms the counter, reads every contributor's slot in workgroup order
// (= k order: deterministic, run-to-run identical) with `sc1` loads and runs the epilogue.  The hand-off is wave-local (no
// __syncthreads, no release/acquire fences: write-through stores + drained vmcnt + relaxed agent-scope ticket, MI355X guide G16) and
// placement-independent.
//
// Two tile shapes: 128 x 128 (4 waves x 64 x 64, as csrc/sgemm.hip) when a workgroup's share is >= 12 such chunks, else 64 x 64 (4 waves
// x 32 x 32): four times the tiles, so a 500-row product is cut into 1-4 slices per tile instead of 8-16, and the partial traffic
// stays a fraction of the output.  f32 MFMA issues one 32x32x2 per 64 cycles: LDS bandwidth is nowhere near a limit at either shape.
// Operand storage, LDS layouts and the chunk pipeline follow csrc/sgemm.hip (two register prefetch sets, one barrier per chunk).
//
// Epilogue extras for the training step: act 2 with C2 != NULL also stores the pre-activation (fc1: hpre for the GELU backward, h for
// the fc2 weight gradient, one launch instead of GEMM + gelu_fwd); act 3 multiplies by GELU'(aux[m][n]) (dX of fc2 -> dhpre, instead
// of GEMM + gelu_bwd).
#include "sfmi_common.h"
#include <mutex>

#define SK_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SK_KC = 32;             // k per chunk
constexpr int SK_KS = SK_KC + 4;      // row stride of a K-contiguous LDS tile
constexpr int SK_GM = 8;              // M-tiles per panel of the tile order

struct SkArgs {
  const float* A; const float* B; float* C; float* C2; const float* bias; const float* aux; const float* resid;
  int M, N, K, lda, ldb, ldc, accumulate, act;
  float drop_p; unsigned drop_seed; const unsigned* drop_seed_dev;   // drop_seed_dev != NULL: the seed is read from device memory (captured training step)
  float* slab;     // [G][2][BM * BN] partial tiles (slot 0: a workgroup's first segment, slot 1: its last)
  int* cnt;        // [tiles][4] arrival tickets, zero before the first launch, re-armed by the last arriver
  int nbm, nbn, nch, stagger;
  long long units; // tiles * nch
};

__device__ __forceinline__ float sk_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float sk_gelu_grad(float x) {      // == csrc/train.hip:gelu_grad
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// Issue order of one pipelined iteration (TM == 1): the 16 MFMAs of a chunk form one dependent chain (64 cycles each), so everything
// else of the iteration - the selects and LDS stores of [A], the address arithmetic and global loads of [B] - is issued in the shadow of
// an MFMA that is already executing: MFMA, a few VALU, one LDS store (first four gaps) or one global load (next four gaps), MFMA, ...
// Serialising [A] / [B] in front of the chain instead costs ~400 issue cycles per 1024 (measured: 0.52 against 0.45 ms per block).
__device__ __forceinline__ void sk_interleave() {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
    __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);      // VALU / SALU that became ready
    if (i < 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        // one LDS store
    else if (i < 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one global load
  }
}

// buffer descriptor over a wave-uniform address (readfirstlane makes the uniformity explicit: no waterfall loop around the buffer ops)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk_rsrc(const float* p, int bytes) {
  const unsigned long long x = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x), hi = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}

// first unit of workgroup v: floor(v * U / G); owner of unit u: the largest v whose first unit is <= u
__device__ __forceinline__ long long sk_first_unit(long long v, long long U, long long G) { return v * U / G; }
__device__ __forceinline__ int sk_owner(long long u, long long U, long long G) { return (int)(((u + 1) * G - 1) / U); }

template <bool AK, bool BK, int TM, int LOOP>      // LOOP (TM == 1 only): 0 = store / barrier / load / read / MFMA per chunk, 1 = pipelined + interleaved
__global__ __launch_bounds__(256, 2) void sgemm_sk_kernel(SkArgs a) {
  constexpr bool PIPE = TM == 1 && LOOP == 1;
  constexpr int BM = 64 * TM, BN = 64 * TM;          // workgroup tile
  constexpr int NL = 2 * TM;                         // float4 loads per thread, operand and chunk
  constexpr int RS = BM + 4;                         // row stride of a row-contiguous LDS tile ([k][rows])
  constexpr int TILE = BM * SK_KS;                   // floats per operand tile in either orientation (32 * (BM + 4) <= BM * 36)
  constexpr int CPR = 16 * TM;                       // float4 per k-row of a row-contiguous tile
  constexpr int NACC = 16 * TM * TM;                 // accumulator floats per lane
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  float* As = sk_lds;                    // [2][TILE]
  float* Bs = sk_lds + 2 * TILE;         // [2][TILE]
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const long long G = gridDim.x, U = a.units;
  // `sk_stagger` experiment: the second resident workgroup of every CU starts half a chunk late, so that the two waves of a SIMD take
  // turns on the matrix pipe instead of colliding on it and idling together (timing only: no result depends on it)
  if (a.stagger > 0 && blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(4);
  long long v = blockIdx.x;
  {    // block b runs on XCD b % 8: give every XCD one contiguous range of the work (bijective for any G)
    const long long q = G / 8, r = G % 8, xcd = v % 8, k = v / 8;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const long long u_lo = sk_first_unit(v, U, G), u_hi = sk_first_unit(v + 1, U, G);
  const int C = a.nch;

  // ---- staging ------------------------------------------------------------------------------------------------------------------
  f32x4 ra[2][NL], rb[2][NL];
  const float* pa[NL];
  const float* pb[NL];
  int ka[NL], kb[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    ka[i] = AK ? 4 * (tid & 7) : tid / CPR + (256 / CPR) * i;
    kb[i] = BK ? 4 * (tid & 7) : tid / CPR + (256 / CPR) * i;
  }
  // Plain loops: the per-thread pointers include the thread's k offset inside a chunk (as in csrc/sgemm.hip).  Pipelined loop: they point at k = 0
  // of the thread's row / column, and the k term is added per load - so that an out-of-range k can fall back to k = 0 (a valid address)
  auto set_tile = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int kpa = !PIPE ? ka[i] : 0, kpb = !PIPE ? kb[i] : 0;
      if (AK) pa[i] = a.A + (long long)min(m0 + (tid >> 3) + 32 * i, a.M - 1) * a.lda + kpa;
      else pa[i] = a.A + (long long)kpa * a.lda + min(m0 + 4 * (tid % CPR), a.M - 4);
      if (BK) pb[i] = a.B + (long long)min(n0 + (tid >> 3) + 32 * i, a.N - 1) * a.ldb + kpb;
      else pb[i] = a.B + (long long)kpb * a.ldb + min(n0 + 4 * (tid % CPR), a.N - 4);
    }
  };
  // Branch-free: a k beyond K (the zero-filled tail of the last chunk) reads a valid address instead (k offset 0) and the VALUE is
  // replaced by zero when the registers are handed to LDS (store_chunk: not here, or the select would wait for the load at once).  A conditional load - or two code paths that join - would make hipcc wait for ALL outstanding loads
  // at the next use (the prefetch of the other register set included), which exposes a full memory latency per chunk.
  auto load_chunk = [&](f32x4 (&xa)[NL], f32x4 (&xb)[NL], int k0) {
    if constexpr (!PIPE) {      // (the 128 x 128 form sits at the 256-register budget: the two-path loads of csrc/sgemm.hip, no select to carry)
      const long long oa = AK ? (long long)k0 : (long long)k0 * a.lda, ob = BK ? (long long)k0 : (long long)k0 * a.ldb;
      if (k0 + SK_KC <= a.K) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          xa[i] = *reinterpret_cast<const f32x4*>(pa[i] + oa);
          xb[i] = *reinterpret_cast<const f32x4*>(pb[i] + ob);
        }
      } else {      // the tail chunk zero-fills k >= K
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          xa[i] = k0 + ka[i] < a.K ? *reinterpret_cast<const f32x4*>(pa[i] + oa) : f32x4{0.f, 0.f, 0.f, 0.f};
          xb[i] = k0 + kb[i] < a.K ? *reinterpret_cast<const f32x4*>(pb[i] + ob) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int kA = k0 + ka[i] < a.K ? k0 + ka[i] : 0, kB = k0 + kb[i] < a.K ? k0 + kb[i] : 0;      // k = 0 .. 3 always exists
        const long long oa = AK ? (long long)kA : (long long)kA * a.lda, ob = BK ? (long long)kB : (long long)kB * a.ldb;
        xa[i] = *reinterpret_cast<const f32x4*>(pa[i] + oa);
        xb[i] = *reinterpret_cast<const f32x4*>(pb[i] + ob);
      }
    }
  };
  auto store_chunk = [&](const f32x4 (&xa)[NL], const f32x4 (&xb)[NL], int buf, int k0) {      // k0: the k the registers were loaded for
    float* as = As + buf * TILE;
    float* bs = Bs + buf * TILE;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const f32x4 va = (!PIPE || k0 + ka[i] < a.K) ? xa[i] : zero, vb = (!PIPE || k0 + kb[i] < a.K) ? xb[i] : zero;
      if (AK) *reinterpret_cast<f32x4*>(as + ((tid >> 3) + 32 * i) * SK_KS + 4 * (tid & 7)) = va;
      else *reinterpret_cast<f32x4*>(as + ka[i] * RS + 4 * (tid % CPR)) = va;
      if (BK) *reinterpret_cast<f32x4*>(bs + ((tid >> 3) + 32 * i) * SK_KS + 4 * (tid & 7)) = vb;
      else *reinterpret_cast<f32x4*>(bs + kb[i] * RS + 4 * (tid % CPR)) = vb;
    }
  };

  f32x16 acc[TM][TM];     // [n tile][m tile]
  // TM == 1: one 32 x 32 tile per wave, 16 MFMAs on ONE accumulator per chunk.  All eight operand fragments of a chunk (32 registers)
  // are read in one burst right after the barrier that publishes the chunk, i.e. one iteration AHEAD of their MFMAs (see the loop).
  f32x4 fm[SK_KC / 8], fn[SK_KC / 8];
  auto read_frags = [&](int buf) {
    const float* as = As + buf * TILE;
    const float* bs = Bs + buf * TILE;
#pragma unroll
    for (int g = 0; g < SK_KC / 8; ++g) {
      if (AK) fm[g] = *reinterpret_cast<const f32x4*>(as + (32 * wm + pl) * SK_KS + 8 * g + 4 * hi);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) fm[g][q] = as[(8 * g + 4 * hi + q) * RS + 32 * wm + pl];
      }
      if (BK) fn[g] = *reinterpret_cast<const f32x4*>(bs + (32 * wn + pl) * SK_KS + 8 * g + 4 * hi);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) fn[g][q] = bs[(8 * g + 4 * hi + q) * RS + 32 * wn + pl];
      }
    }
  };
  auto mfma_frags = [&]() {
#pragma unroll
    for (int g = 0; g < SK_KC / 8; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[0][0] = SK_MFMA(fn[g][q], fm[g][q], acc[0][0]);
  };

  auto compute = [&](int buf) {
    const float* as = As + buf * TILE;
    const float* bs = Bs + buf * TILE;
    if constexpr (TM == 1) {
      (void)as; (void)bs;
      read_frags(buf);                        // all eight fragments of the chunk up front (the compiler would read 2, wait, issue 4 MFMAs)
      __builtin_amdgcn_sched_barrier(0);
      mfma_frags();
    } else {
#pragma unroll
      for (int g = 0; g < SK_KC / 8; ++g) {     // k8 groups: MFMA q of the group multiplies k = 8 g + 4 hi + q
        f32x4 mf[TM], nf[TM];
        if (AK) {
#pragma unroll
          for (int j = 0; j < TM; ++j) mf[j] = *reinterpret_cast<const f32x4*>(as + (64 * wm + 32 * j + pl) * SK_KS + 8 * g + 4 * hi);
        } else {    // tile t = rows 2 pl + t (interleaved): one ds_read_b64 per k feeds both tiles; the epilogue undoes it
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(as + (8 * g + 4 * hi + q) * RS + 64 * wm + 2 * pl);
            mf[0][q] = t[0]; mf[TM - 1][q] = t[1];
          }
        }
        if (BK) {
#pragma unroll
          for (int i = 0; i < TM; ++i) nf[i] = *reinterpret_cast<const f32x4*>(bs + (64 * wn + 32 * i + pl) * SK_KS + 8 * g + 4 * hi);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(bs + (8 * g + 4 * hi + q) * RS + 64 * wn + 2 * pl);
            nf[0][q] = t[0]; nf[TM - 1][q] = t[1];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = SK_MFMA(nf[i][q], mf[j][q], acc[i][j]);
      }
    }
  };

  // ---- epilogue of one finished tile (this wave's sub-tile) ---------------------------------------------------------------------
  auto emit = [&](int m, int n, f32x4 x) {
    if (m >= a.M || n >= a.N) return;
    const long long off = (long long)m * a.ldc + n;
    float* cp = a.C + off;
    if (a.accumulate) x = x + *reinterpret_cast<const f32x4*>(cp);
    if (a.bias) x = x + *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
    } else if (a.act == 2) {
      if (a.C2) *reinterpret_cast<f32x4*>(a.C2 + off) = x;
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = sk_gelu(x[e]);
    } else if (a.act == 3) {
      const f32x4 h = *reinterpret_cast<const f32x4*>(a.aux + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] *= sk_gelu_grad(h[e]);
    }
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] *= sfmi_dropout_mul(a.drop_seed_dev ? *a.drop_seed_dev : a.drop_seed, (unsigned)(m * a.N + n + e), a.drop_p, 1.0f / (1.0f - a.drop_p));
    }
    if (a.resid) x = x + *reinterpret_cast<const f32x4*>(a.resid + off);
    *reinterpret_cast<f32x4*>(cp) = x;
  };
  auto epilogue = [&](int m0, int n0) {
    // lane (pl, hi), register 4 gg + r of tile (i, j) is output (m tile-row pl, n tile-row 8 gg + 4 hi + r)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + 32 * TM * wm + ((AK || TM == 1) ? 32 * j + pl : 2 * pl + j);
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        if (BK || TM == 1) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            emit(m, n0 + 32 * TM * wn + 32 * i + 8 * gg + 4 * hi, f32x4{acc[i][j][4 * gg], acc[i][j][4 * gg + 1], acc[i][j][4 * gg + 2], acc[i][j][4 * gg + 3]});
        } else {   // n = 2 (8 gg + 4 hi + r) + i: the two n tiles interleave into 8 consecutive columns per lane
          const int n = n0 + 64 * wn + 16 * gg + 8 * hi;
          emit(m, n, f32x4{acc[0][j][4 * gg], acc[TM - 1][j][4 * gg], acc[0][j][4 * gg + 1], acc[TM - 1][j][4 * gg + 1]});
          emit(m, n + 4, f32x4{acc[0][j][4 * gg + 2], acc[TM - 1][j][4 * gg + 2], acc[0][j][4 * gg + 3], acc[TM - 1][j][4 * gg + 3]});
        }
      }
    }
  };

  // ---- the workgroup's segments ------------------------------------------------------------------------------------------------
  const int tile_lo = (int)(u_lo / C);
  for (long long u = u_lo; u < u_hi;) {
    const int tile = (int)(u / C);
    const int c_lo = (int)(u - (long long)tile * C);
    const int c_hi = (int)min((long long)C, c_lo + (u_hi - u));
    const int nchunks = c_hi - c_lo;
    int mt, ntl;
    {   // panels of SK_GM M-tiles; inside a panel the M-tiles of one N-tile are consecutive (operand tiles shared through the XCD's L2)
      const long long per_panel = (long long)SK_GM * a.nbn;
      const int panel = (int)(tile / per_panel);
      const int rows_in_panel = min(SK_GM, a.nbm - panel * SK_GM);
      const long long r = tile - panel * per_panel;
      ntl = (int)(r / rows_in_panel);
      mt = panel * SK_GM + (int)(r % rows_in_panel);
    }
    const int m0 = mt * BM, n0 = ntl * BN;
    set_tile(m0, n0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.0f;
    load_chunk(ra[0], rb[0], c_lo * SK_KC);
    if (PIPE) __builtin_amdgcn_sched_barrier(0);      // keep the prologue's loads in chunk order (see the loop's comment)
    if (PIPE || nchunks > 1) load_chunk(ra[1], rb[1], min(c_lo + 1, c_lo + nchunks - 1) * SK_KC);
    if constexpr (PIPE) {
      // Pipelined so that a wave's MFMAs run back to back (PMC of the plain loop at 500 rows: matrix pipes busy 45 % of the launch, the
      // waves of a workgroup - and of the two workgroups of a CU - hit store / barrier / fragment-read phases together).  Iteration c:
      //   [A] chunk c+1: staging registers -> LDS buffer (c+1)&1     (its global loads were issued two iterations ago)
      //   [B] chunk c+3: global loads into the registers [A] just freed
      //   [C] the 16 MFMAs of chunk c, from fragment registers read in the PREVIOUS iteration
      //   [D] barrier: chunk c+1 is published (every wave's reads of that buffer - [E] of iteration c-1 - precede it in program order)
      //   [E] the fragments of chunk c+1 -> registers
      // The LDS-write latency of [A] and the read latency of [E] run under the MFMAs instead of between them; one barrier per chunk.
      // The body carries no condition but the loop exit: loads beyond the segment re-read its last chunk, stores / fragment reads
      // beyond it move unused data through a free buffer - so the compiler counts the loads in flight exactly (vmcnt(4): the
      // other register set's prefetch stays in flight across the use of this one).
      // The loads must also be ISSUED in chunk order, in the prologue exactly as in the body (sched_barrier between the groups): the
      // wait-count pass merges the loop-entry state with the back edge, and a prologue whose loads were interleaved (hipcc groups them by
      // address) degrades every wait of the steady state to "all loads done".
      const int last = c_lo + nchunks - 1;
      __builtin_amdgcn_sched_barrier(0);
      store_chunk(ra[0], rb[0], 0, c_lo * SK_KC);
      __builtin_amdgcn_sched_barrier(0);
      load_chunk(ra[0], rb[0], min(c_lo + 2, last) * SK_KC);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      read_frags(0);
      __builtin_amdgcn_sched_barrier(0);
      // No exit in the middle of the body either (an odd last chunk is peeled off below): the structurised back edge would merge
      // the state of the early exit into the loop header and degrade the waits of the first half again.
      for (int c = 0; c + 1 < nchunks; c += 2) {
        store_chunk(ra[1], rb[1], 1, min(c_lo + c + 1, last) * SK_KC);       // [A] chunk c+1
        load_chunk(ra[1], rb[1], min(c_lo + c + 3, last) * SK_KC);           // [B] chunk c+3
        mfma_frags();                                                        // [C] chunk c
        sk_interleave();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                                     // [D]
        read_frags(1);                                                       // [E] chunk c+1
        __builtin_amdgcn_sched_barrier(0);
        store_chunk(ra[0], rb[0], 0, min(c_lo + c + 2, last) * SK_KC);       // [A] chunk c+2
        load_chunk(ra[0], rb[0], min(c_lo + c + 4, last) * SK_KC);           // [B] chunk c+4
        mfma_frags();                                                        // [C] chunk c+1
        sk_interleave();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_frags(0);                                                       // [E] chunk c+2
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nchunks & 1) mfma_frags();      // the last chunk of an odd count: its fragments were read by the prologue / the last [E]
    } else {
      for (int c = 0; c < nchunks; c += 2) {
        store_chunk(ra[0], rb[0], 0, 0);
        __syncthreads();
        if (c + 2 < nchunks) load_chunk(ra[0], rb[0], (c_lo + c + 2) * SK_KC);
        compute(0);
        if (c + 1 < nchunks) {
          store_chunk(ra[1], rb[1], 1, 0);
          __syncthreads();
          if (c + 3 < nchunks) load_chunk(ra[1], rb[1], (c_lo + c + 3) * SK_KC);
          compute(1);
        }
      }
    }
    bool finished = nchunks == C;      // wave-uniform
    if (!finished) {
      // a slice of a tile: publish this wave's accumulators write-through into the workgroup's slot, drain, take a ticket
      const int v_first = sk_owner((long long)tile * C, U, G), v_last = sk_owner((long long)tile * C + C - 1, U, G);
      const int slot = tile == tile_lo ? 0 : 1;
      {
        const __amdgpu_buffer_rsrc_t rs = sk_rsrc(a.slab + (((long long)v * 2 + slot) * 4 + wave) * (64 * NACC), 64 * NACC * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
              const u32x4 w = {__float_as_uint(acc[i][j][4 * gg]), __float_as_uint(acc[i][j][4 * gg + 1]), __float_as_uint(acc[i][j][4 * gg + 2]),
                               __float_as_uint(acc[i][j][4 * gg + 3])};
              __builtin_amdgcn_raw_buffer_store_b128(w, rs, (((i * TM + j) * 4 + gg) * 64 + lane) * 16, 0, /*sc1*/ 16);
            }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int ticket = 0;
      int* cnt = a.cnt + (long long)tile * 4 + wave;
      if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if (ticket == v_last - v_first) {      // every other slice of this (tile, wave) has been published: add them in k order
        if (lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next launch
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.0f;
        // workgroup w's slot for this tile: 0 when the tile is where w's range starts, else 1 (its last segment)
        auto slot_rsrc = [&](int w) {
          const int wslot = (int)(sk_first_unit(w, U, G) / C) == tile ? 0 : 1;
          return sk_rsrc(a.slab + (((long long)w * 2 + wslot) * 4 + wave) * (64 * NACC), 64 * NACC * 4);
        };
        if constexpr (TM == 1) {
          // the slices of up to four contributors are requested together (16 registers each would be 64 for the 128 x 128 form: that one
          // stays at one contributor per round trip): a serial round trip per contributor was most of a cut tile's tail
          for (int w0 = v_first; w0 <= v_last; w0 += 4) {
            u32x4 t[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const __amdgpu_buffer_rsrc_t rs = slot_rsrc(min(w0 + k, v_last));
#pragma unroll
              for (int x = 0; x < 4; ++x) t[k][x] = __builtin_amdgcn_raw_buffer_load_b128(rs, (x * 64 + lane) * 16, 0, /*sc1*/ 16);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (w0 + k > v_last) break;        // added in k order (deterministic)
#pragma unroll
              for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][0][4 * gg + e] += __uint_as_float(t[k][gg][e]);
            }
          }
        } else {
          for (int w = v_first; w <= v_last; ++w) {
            const __amdgpu_buffer_rsrc_t rs = slot_rsrc(w);
            u32x4 t[TM * TM * 4];
#pragma unroll
            for (int x = 0; x < TM * TM * 4; ++x) t[x] = __builtin_amdgcn_raw_buffer_load_b128(rs, (x * 64 + lane) * 16, 0, /*sc1*/ 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                  for (int e = 0; e < 4; ++e) acc[i][j][4 * gg + e] += __uint_as_float(t[(i * TM + j) * 4 + gg][e]);
          }
        }
        finished = true;
      }
    }
    if (finished) epilogue(m0, n0);
    u += nchunks;
    if (u < u_hi) __syncthreads();      // the next segment's first store_chunk rewrites LDS tiles other waves may still be reading
  }
}

template <bool AK, bool BK, int TM, int LOOP>
hipError_t sk_attr() {
  return hipFuncSetAttribute((const void*)sgemm_sk_kernel<AK, BK, TM, LOOP>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * TM * SK_KS * (int)sizeof(float));
}

}  // namespace

extern "C" {

// Tile shape of sfmi_sgemm_sk_f32 for a product: 2 = 128 x 128 workgroup tiles, 1 = 64 x 64 (the `sk_tile` knob overrides).
int sfmi_sgemm_sk_tile(int M, int N, int K) {
  (void)K;
  if (g_sfmi_tune.sk_tile) return g_sfmi_tune.sk_tile;
  // 128 x 128 tiles run ~12 % faster per FLOP (half the operand traffic and barriers) but a cut tile costs a 64 KB slab round trip:
  // they pay once a workgroup's share is >= 12 chunks of 128 x 128 x 32 (profiles/r05_kbench_sk_call1.txt: 3992-row products and
  // long-K weight gradients); the 500-row products of the batch-1 step stay on 64 x 64
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128), nch = (K + 31) / 32;
  return t128 * nch >= 12 * 512 ? 2 : 1;
}
// Scratch of sfmi_sgemm_sk_f32, valid for ANY product and any `sk_grid`: slab floats (1024 workgroups x 2 slots x 128 x 128 = 134 MB)
// and the ticket counters a product of this output size needs (4 per 64 x 64 tile; zero them ONCE, the kernel re-arms them).
long long sfmi_sgemm_sk_slab_floats(void) { return 1024ll * 2 * 128 * 128; }
long long sfmi_sgemm_sk_cnt_ints(int M, int N) { return 4ll * ((M + 63) / 64) * ((N + 63) / 64); }

// Row-major C (M,N; ldc) = op(A) op(B) (+ C when accumulate) (+ bias[n]) -> act -> dropout -> (+ resid (M,N; ldc)); operand forms as
// sfmi_sgemm_mfma_f32.  act: 0 none, 1 ReLU, 2 GELU(erf) (C2 != NULL also receives the pre-activation, (M,N; ldc)), 3 multiply by
// GELU'(aux[m][n]) (aux (M,N; ldc)).  slab / cnt: caller-owned scratch (sfmi_sgemm_sk_slab_floats / _cnt_ints; cnt zeroed once), one
// pair per stream that may run these launches concurrently.  Deterministic: a tile's slices are added in k order.
// Replaces the cuBLAS sgemm behind nn.Linear and its autograd (mingpt.py:46-111) in the training step at small batch.
int sfmi_sgemm_sk_sd_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, float* C2,
                         int ldc, int accumulate, const float* bias, int act, const float* aux, const float* resid, float drop_p,
                         unsigned drop_seed, const unsigned* drop_seed_dev, float* slab, long long slab_floats, int* cnt, long long cnt_ints,
                         void* stream);
int sfmi_sgemm_sk_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, float* C2,
                      int ldc, int accumulate, const float* bias, int act, const float* aux, const float* resid, float drop_p,
                      unsigned drop_seed, float* slab, long long slab_floats, int* cnt, long long cnt_ints, void* stream) {
  return sfmi_sgemm_sk_sd_f32(transA, transB, M, N, K, A, lda, B, ldb, C, C2, ldc, accumulate, bias, act, aux, resid, drop_p, drop_seed, nullptr, slab,
                              slab_floats, cnt, cnt_ints, stream);
}
// the same with the dropout seed optionally in device memory (drop_seed_dev != NULL overrides drop_seed at run time)
int sfmi_sgemm_sk_sd_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, float* C2,
                         int ldc, int accumulate, const float* bias, int act, const float* aux, const float* resid, float drop_p,
                         unsigned drop_seed, const unsigned* drop_seed_dev, float* slab, long long slab_floats, int* cnt, long long cnt_ints,
                         void* stream) {
  if (!A || !B || !C || !slab || !cnt || M <= 0 || N <= 0 || K <= 0 || N % 4 || lda % 4 || ldb % 4 || ldc % 4) return SFMI_EINVAL;
  if (!transA && K % 4) return SFMI_EINVAL;          // A K-contiguous: float4 along k
  if (transA && (M % 4 || M < 4)) return SFMI_EINVAL; // A row-contiguous: float4 along m
  if (transB && K % 4) return SFMI_EINVAL;
  if (N < 4 || act < 0 || act > 3 || (act == 3 && !aux) || drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  const int TM = sfmi_sgemm_sk_tile(M, N, K), BM = 64 * TM;
  SkArgs a;
  a.A = A; a.B = B; a.C = C; a.C2 = C2; a.bias = bias; a.aux = aux; a.resid = resid; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb;
  a.ldc = ldc; a.accumulate = accumulate; a.act = act; a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_seed_dev = drop_seed_dev; a.slab = slab; a.cnt = cnt;
  a.nbm = (M + BM - 1) / BM; a.nbn = (N + BM - 1) / BM; a.nch = (K + SK_KC - 1) / SK_KC;
  const long long tiles = (long long)a.nbm * a.nbn;
  a.units = tiles * a.nch;
  if (tiles > 0x3fffffffLL || tiles * 4 > cnt_ints) return SFMI_EINVAL;
  long long G = g_sfmi_tune.sk_grid;                   // default 512: two resident workgroups per CU
  if (G > a.units) G = a.units;
  if (G * 2 * BM * BM > slab_floats) return SFMI_EINVAL;
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [] {
    const hipError_t es[12] = {sk_attr<true, true, 1, 0>(), sk_attr<true, false, 1, 0>(), sk_attr<false, false, 1, 0>(), sk_attr<false, true, 1, 0>(),
                               sk_attr<true, true, 1, 1>(), sk_attr<true, false, 1, 1>(), sk_attr<false, false, 1, 1>(), sk_attr<false, true, 1, 1>(),
                               sk_attr<true, true, 2, 0>(), sk_attr<true, false, 2, 0>(), sk_attr<false, false, 2, 0>(), sk_attr<false, true, 2, 0>()};
    for (hipError_t e : es)
      if (e != hipSuccess) attr_err = e;
  });
  if (attr_err != hipSuccess) return SFMI_ELDS;
  const size_t lds = (size_t)4 * BM * SK_KS * sizeof(float);
  const dim3 grid((unsigned)G), block(256);
  hipStream_t st = (hipStream_t)stream;
  a.stagger = g_sfmi_tune.sk_stagger;
  const int loop = g_sfmi_tune.sk_loop;
#define SK_LAUNCH(AK_, BK_) do { \
    if (TM == 2) hipLaunchKernelGGL((sgemm_sk_kernel<AK_, BK_, 2, 0>), grid, block, lds, st, a); \
    else if (loop == 1) hipLaunchKernelGGL((sgemm_sk_kernel<AK_, BK_, 1, 1>), grid, block, lds, st, a); \
    else hipLaunchKernelGGL((sgemm_sk_kernel<AK_, BK_, 1, 0>), grid, block, lds, st, a); } while (0)
  if (!transA && transB) SK_LAUNCH(true, true);
  else if (!transA && !transB) SK_LAUNCH(true, false);
  else if (transA && !transB) SK_LAUNCH(false, false);
  else SK_LAUNCH(false, true);
#undef SK_LAUNCH
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
