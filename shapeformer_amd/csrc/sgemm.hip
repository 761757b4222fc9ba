// Plain f32 GEMM on the matrix cores (gfx950, v_mfma_f32_32x32x2_f32 - exact f32): the large products of the transformer's
// prefill and training step (mingpt.py:46-111 Linear layers at M = B*L rows, and their autograd: dX = dY W, dW = dY^T X).
//
//   C (M,N; ldc) = op(A) (M,K) * op(B) (K,N)  [+ C]  [+ bias[n]] [act] [+ resid]
//
// Operand storage is a template choice, so that none of the three forms needs an explicit transpose pass:
//   AK = true : A stored (M,K), K contiguous           AK = false: A stored (K,M), M contiguous   (A = dY^T for dW)
//   BK = true : B stored (N,K), K contiguous (nn.Linear weight: y = x W^T)   BK = false: B stored (K,N), N contiguous
// Tile 128 x 128 x 32 per workgroup (4 waves, each 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator registers), two workgroups
// per CU; LDS double-buffered, the global loads of chunk c+1 are in flight (registers) under the 64 MFMAs per wave of
// chunk c; ONE barrier per 32-deep chunk.  LDS tiles keep the SOURCE orientation (no transposing stores):
//   K-contiguous operand -> [row][32 + 4] floats, fragments read as ds_read_b128 (row stride 36: the 16-lane groups of a
//   b128 read hit 16 distinct bank quads);  row-contiguous operand -> [k][128 + 4] floats, fragments read as ds_read_b32
//   (32 consecutive floats per half-wave).
// (Tried and dropped, round 2: the same tiles brought in by LDS-DMA `global_load_lds_dwordx4` with an XOR-swizzled unpadded
//  layout, all fragments of a chunk read before the next chunk's DMA is issued - correct, but 105-115 TFLOP/s against 113-128
//  for the register-staged loop below: the fragment burst no longer overlaps the MFMAs.)
// Block order is XCD-aware (block b runs on XCD b % 8): the N-tiles of one M-tile are consecutive on ONE XCD, so the A tile is
// fetched from HBM once per XCD L2 instead of once per N-tile.
#include "sfmi_common.h"
#include <mutex>

#define SG_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int SG_BM = 128, SG_BN = 128, SG_KC = 32;
constexpr int SG_GM = 8;                // M-tiles per panel of the block order
constexpr int SG_KS = SG_KC + 4;      // row stride of a K-contiguous LDS tile
constexpr int SG_RS = SG_BM + 4;      // row stride of a row-contiguous LDS tile ([k][rows])
constexpr int SG_TILE = 128 * 36;     // floats per operand tile in either orientation (32 * 132 = 4224 <= 4608)

struct SgemmArgs {
  const float* A; const float* B; float* C; const float* bias; const float* resid;
  int M, N, K, lda, ldb, ldc, accumulate, act;
  float drop_p; unsigned drop_seed;   // nn.Dropout on the product (after bias / activation, before the residual): mingpt.py:90,105
  float* ws;      // split-K (gridDim.y > 1): split s writes its partial product to ws + s*M*N (row stride N), no epilogue
};

template <bool AK, bool BK>
__global__ __launch_bounds__(256, 2) void sgemm_mfma_kernel(SgemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sg_lds[];
  float* As = sg_lds;                    // [2][SG_TILE]
  float* Bs = sg_lds + 2 * SG_TILE;      // [2][SG_TILE]
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const int nbn = (a.N + SG_BN - 1) / SG_BN;
  const long long nb = (long long)gridDim.x;
  long long lid = blockIdx.x;
  {
    const long long q = nb / 8, r = nb % 8, xcd = lid % 8, k = lid / 8;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any nb
  }
  // within the XCD's contiguous range the blocks walk PANELS of SG_GM M-tiles: for every N-tile the panel's M-tiles are
  // consecutive, so the ~64 blocks resident on an XCD at a time cover ~8 M-tiles x ~8 N-tiles (each A and each B tile is
  // shared by 8 resident blocks through the XCD's L2) instead of 2 M-tiles x all N-tiles
  int mt, ntl;
  {
    const int nbm = (a.M + SG_BM - 1) / SG_BM;
    const long long per_panel = (long long)SG_GM * nbn;
    const int panel = (int)(lid / per_panel);
    const int rows_in_panel = min(SG_GM, nbm - panel * SG_GM);
    const long long r = lid - panel * per_panel;
    ntl = (int)(r / rows_in_panel);
    mt = panel * SG_GM + (int)(r % rows_in_panel);
  }
  const int m0 = mt * SG_BM, n0 = ntl * SG_BN;

  // ---- staging: 4 float4 of A and 4 float4 of B per thread and chunk ---------------------------------------------------
  f32x4 ra[2][4], rb[2][4];      // two prefetch sets: the loads of chunk c+2 are issued while chunk c computes
  // Per-thread source pointers are fixed for the whole K loop (rows are clamped: their results are masked in the epilogue); a
  // chunk that lies fully inside K - every chunk but possibly the last - loads with no per-element condition (wave-uniform
  // branch), the tail chunk zero-fills k >= K.  (Per-load guards cost ~10 % of the kernel: round-2 ablation, profiles/r02_decode_step_experiments.md section 3.)
  const float* pa[4];
  const float* pb[4];
  int ka[4], kb[4];      // k offset of the element inside a chunk
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (AK) { ka[i] = 4 * (tid & 7); pa[i] = a.A + (long long)min(m0 + (tid >> 3) + 32 * i, a.M - 1) * a.lda + ka[i]; }
    else { ka[i] = (tid >> 5) + 8 * i; pa[i] = a.A + (long long)ka[i] * a.lda + min(m0 + 4 * (tid & 31), a.M - 4); }
    if (BK) { kb[i] = 4 * (tid & 7); pb[i] = a.B + (long long)min(n0 + (tid >> 3) + 32 * i, a.N - 1) * a.ldb + kb[i]; }
    else { kb[i] = (tid >> 5) + 8 * i; pb[i] = a.B + (long long)kb[i] * a.ldb + min(n0 + 4 * (tid & 31), a.N - 4); }
  }
  auto load_chunk = [&](f32x4 (&ra)[4], f32x4 (&rb)[4], int k0) {
    const long long oa = AK ? (long long)k0 : (long long)k0 * a.lda, ob = BK ? (long long)k0 : (long long)k0 * a.ldb;
    if (k0 + SG_KC <= a.K) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(pa[i] + oa);
        rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + ob);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = k0 + ka[i] < a.K ? *reinterpret_cast<const f32x4*>(pa[i] + oa) : f32x4{0.f, 0.f, 0.f, 0.f};
        rb[i] = k0 + kb[i] < a.K ? *reinterpret_cast<const f32x4*>(pb[i] + ob) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto store_chunk = [&](const f32x4 (&ra)[4], const f32x4 (&rb)[4], int buf) {
    float* as = As + buf * SG_TILE;
    float* bs = Bs + buf * SG_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (AK) *reinterpret_cast<f32x4*>(as + ((tid >> 3) + 32 * i) * SG_KS + 4 * (tid & 7)) = ra[i];
      else *reinterpret_cast<f32x4*>(as + ((tid >> 5) + 8 * i) * SG_RS + 4 * (tid & 31)) = ra[i];
      if (BK) *reinterpret_cast<f32x4*>(bs + ((tid >> 3) + 32 * i) * SG_KS + 4 * (tid & 7)) = rb[i];
      else *reinterpret_cast<f32x4*>(bs + ((tid >> 5) + 8 * i) * SG_RS + 4 * (tid & 31)) = rb[i];
    }
  };

  f32x16 acc[2][2];     // [n tile][m tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.0f;

  const int nch_all = (a.K + SG_KC - 1) / SG_KC, S = gridDim.y, sp = blockIdx.y;
  const int c_lo = (int)((long long)nch_all * sp / S), c_hi = (int)((long long)nch_all * (sp + 1) / S);
  const int nchunks = c_hi - c_lo;
  auto compute = [&](int buf) {
    const float* as = As + buf * SG_TILE;
    const float* bs = Bs + buf * SG_TILE;
#pragma unroll
    for (int g = 0; g < SG_KC / 8; ++g) {     // k8 groups: MFMA q of the group multiplies k = 8 g + 4 hi + q
      f32x4 mf[2], nf[2];
      // K-contiguous tile: tile t of the wave = rows 32 t + pl, one ds_read_b128 per tile and k8 group.
      // Row-contiguous tile ([k][rows]): tile t = rows 2 pl + t (INTERLEAVED), so ONE ds_read_b64 per k yields the operand of
      // both tiles (4 reads per group instead of 8 ds_read_b32); the epilogue undoes the permutation.
      if (AK) {
#pragma unroll
        for (int j = 0; j < 2; ++j) mf[j] = *reinterpret_cast<const f32x4*>(as + (64 * wm + 32 * j + pl) * SG_KS + 8 * g + 4 * hi);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 t = *reinterpret_cast<const f32x2*>(as + (8 * g + 4 * hi + q) * SG_RS + 64 * wm + 2 * pl);
          mf[0][q] = t[0]; mf[1][q] = t[1];
        }
      }
      if (BK) {
#pragma unroll
        for (int i = 0; i < 2; ++i) nf[i] = *reinterpret_cast<const f32x4*>(bs + (64 * wn + 32 * i + pl) * SG_KS + 8 * g + 4 * hi);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 t = *reinterpret_cast<const f32x2*>(bs + (8 * g + 4 * hi + q) * SG_RS + 64 * wn + 2 * pl);
          nf[0][q] = t[0]; nf[1][q] = t[1];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = SG_MFMA(nf[i][q], mf[j][q], acc[i][j]);
    }
  };
  load_chunk(ra[0], rb[0], c_lo * SG_KC);
  if (nchunks > 1) load_chunk(ra[1], rb[1], (c_lo + 1) * SG_KC);
  for (int c = 0; c < nchunks; c += 2) {
    store_chunk(ra[0], rb[0], 0);
    __syncthreads();
    if (c + 2 < nchunks) load_chunk(ra[0], rb[0], (c_lo + c + 2) * SG_KC);
    compute(0);
    if (c + 1 < nchunks) {
      store_chunk(ra[1], rb[1], 1);
      __syncthreads();
      if (c + 3 < nchunks) load_chunk(ra[1], rb[1], (c_lo + c + 3) * SG_KC);
      compute(1);
    }
  }

  // ---- epilogue: lane (pl, hi), register 4 gg + r of tile (i, j) is output (m tile-row pl, n tile-row 8 gg + 4 hi + r) ---------
  const int S_ = S;
  auto emit = [&](int m, int n, f32x4 v) {
    if (m >= a.M || n >= a.N) return;
    if (S_ > 1) {
      *reinterpret_cast<f32x4*>(a.ws + ((long long)sp * a.M + m) * a.N + n) = v;
      return;
    }
    float* cp = a.C + (long long)m * a.ldc + n;
    if (a.accumulate) v = v + *reinterpret_cast<const f32x4*>(cp);
    if (a.bias) v = v + *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.act == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (a.act == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
    }
    if (a.drop_p > 0.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= sfmi_dropout_mul(a.drop_seed, (unsigned)(m * a.N + n + e), a.drop_p, 1.0f / (1.0f - a.drop_p));
    }
    if (a.resid) v = v + *reinterpret_cast<const f32x4*>(a.resid + (long long)m * a.ldc + n);
    *reinterpret_cast<f32x4*>(cp) = v;
  };
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + 64 * wm + (AK ? 32 * j + pl : 2 * pl + j);
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      if (BK) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          emit(m, n0 + 64 * wn + 32 * i + 8 * gg + 4 * hi, f32x4{acc[i][j][4 * gg], acc[i][j][4 * gg + 1], acc[i][j][4 * gg + 2], acc[i][j][4 * gg + 3]});
      } else {   // n = 2 (8 gg + 4 hi + r) + i: the two n tiles interleave into 8 consecutive columns per lane
        const int n = n0 + 64 * wn + 16 * gg + 8 * hi;
        emit(m, n, f32x4{acc[0][j][4 * gg], acc[1][j][4 * gg], acc[0][j][4 * gg + 1], acc[1][j][4 * gg + 1]});
        emit(m, n + 4, f32x4{acc[0][j][4 * gg + 2], acc[1][j][4 * gg + 2], acc[0][j][4 * gg + 3], acc[1][j][4 * gg + 3]});
      }
    }
  }
}

// split-K tail: C = sum_s ws[s] (fixed order: deterministic) (+ C) (+ bias) -> act (+ resid)
__global__ void sgemm_splitk_reduce_kernel(SgemmArgs a, int S) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, n4 = a.N / 4;
  if (i >= (long long)a.M * n4) return;
  const int m = (int)(i / n4), n = (int)(i % n4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(a.ws + (long long)m * a.N + n);
  for (int s = 1; s < S; ++s) v = v + *reinterpret_cast<const f32x4*>(a.ws + ((long long)s * a.M + m) * a.N + n);
  float* cp = a.C + (long long)m * a.ldc + n;
  if (a.accumulate) v = v + *reinterpret_cast<const f32x4*>(cp);
  if (a.bias) v = v + *reinterpret_cast<const f32x4*>(a.bias + n);
  if (a.act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (a.act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
  }
  if (a.drop_p > 0.f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= sfmi_dropout_mul(a.drop_seed, (unsigned)(m * a.N + n + e), a.drop_p, 1.0f / (1.0f - a.drop_p));
  }
  if (a.resid) v = v + *reinterpret_cast<const f32x4*>(a.resid + (long long)m * a.ldc + n);
  *reinterpret_cast<f32x4*>(cp) = v;
}

}  // namespace

extern "C" {

// Row-major C (M,N; ldc) = op(A) op(B) (+ C when accumulate) (+ bias[n]) -> act (0 none, 1 ReLU, 2 GELU(erf)) (+ resid (M,N; ldc)).
// transA == 0: A stored (M,K; lda)   transA != 0: A stored (K,M; lda)      transB == 0: B stored (K,N; ldb)
// transB != 0: B stored (N,K; ldb) (nn.Linear weight).  N, K (and M when transA) multiples of 4; pointers 16-byte aligned.
// Replaces the cuBLAS sgemm behind nn.Linear and its autograd (mingpt.py:46-111) for the prefill / training step.
// Split-K: when the output has too few 128 x 128 tiles to fill the chip (weight gradients: N x K outputs over a deep M), the K
// range is cut into S slices whose partial products go to `ws` (>= S*M*N floats, caller-owned) and are summed in slice order by
// a second small launch; S = sfmi_sgemm_mfma_splits(M, N, K) (1 when ws is NULL / too small).
int sfmi_sgemm_mfma_splits(int M, int N, int K) {
  const long long blocks = (long long)((M + SG_BM - 1) / SG_BM) * ((N + SG_BN - 1) / SG_BN);
  const int nch = (K + SG_KC - 1) / SG_KC;
  int S = 1;
  while (S < 8 && blocks * S * 2 <= 512 && nch / (S * 2) >= 8) S *= 2;   // 512 = 256 CUs x 2 resident workgroups
  return S;
}
int sfmi_sgemm_mfma_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                        int ldc, int accumulate, const float* bias, int act, const float* resid, float* ws, long long ws_floats,
                        float drop_p, unsigned drop_seed, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || N % 4 || lda % 4 || ldb % 4 || ldc % 4) return SFMI_EINVAL;
  if (!transA && K % 4) return SFMI_EINVAL;          // A K-contiguous: float4 along k
  if (transA && (M % 4 || M < 4)) return SFMI_EINVAL; // A row-contiguous: float4 along m
  if (transB && K % 4) return SFMI_EINVAL;
  if (N < 4) return SFMI_EINVAL;
  SgemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.resid = resid; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.accumulate = accumulate; a.act = act; a.ws = ws; a.drop_p = drop_p; a.drop_seed = drop_seed;
  if (drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  int S = ws ? sfmi_sgemm_mfma_splits(M, N, K) : 1;
  while (S > 1 && (long long)S * M * N > ws_floats) S /= 2;
  const long long blocks = (long long)((M + SG_BM - 1) / SG_BM) * ((N + SG_BN - 1) / SG_BN);
  if (blocks > 0x7fffffffLL) return SFMI_EINVAL;
  const size_t lds = 4 * SG_TILE * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  // the 73.7 KB tile pair needs the dynamic-LDS limit raised above the 64 KB default: once per process, and a refusal is an
  // error the caller sees (SFMI_ELDS) instead of a generic launch failure on every prefill / training GEMM
  static std::once_flag once;
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [lds] {
    const void* ks[4] = {(const void*)sgemm_mfma_kernel<true, true>, (const void*)sgemm_mfma_kernel<true, false>,
                         (const void*)sgemm_mfma_kernel<false, false>, (const void*)sgemm_mfma_kernel<false, true>};
    for (const void* k : ks) {
      const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) attr_err = e;
    }
  });
  if (attr_err != hipSuccess) return SFMI_ELDS;
  const dim3 grid((unsigned)blocks, S), block(256);
  if (!transA && transB) hipLaunchKernelGGL((sgemm_mfma_kernel<true, true>), grid, block, lds, st, a);
  else if (!transA && !transB) hipLaunchKernelGGL((sgemm_mfma_kernel<true, false>), grid, block, lds, st, a);
  else if (transA && !transB) hipLaunchKernelGGL((sgemm_mfma_kernel<false, false>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((sgemm_mfma_kernel<false, true>), grid, block, lds, st, a);
  if (S > 1) {
    const long long total = (long long)M * (N / 4);
    hipLaunchKernelGGL(sgemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, S);
  }
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
