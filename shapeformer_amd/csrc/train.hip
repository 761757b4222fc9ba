// Training-step kernels for CondTupleGPT on gfx950 (backward of the blocks, AdamW) — SURVEY §8 a25.
//
// Reference: ShapeFormer.forward / shared_step / configure_optimizers (shapeformer/models/shapeformer/
// shapeformer.py:26-46,132-207), Block / CausalSelfAttention (transformer/mingpt.py:46-111).  The reference gets its
// backward from autograd over ~40 ATen ops per block; here the matmul-shaped gradients reuse the f32-MFMA GEMM
// (sfmi_gemm_f32 on transposed operands) and this file provides everything that is not a GEMM: transposes, column
// reductions, GELU / LayerNorm / softmax-cross-entropy backward, causal-attention backward (recompute form, no LxL
// tensor is stored), embedding-gradient scatter, AdamW.  All reductions have a fixed order or use 2^-32 fixed-point
// integer atomics, so gradients are bit-reproducible run to run.
#include "sfmi_common.h"

// ---------------------------------------------------------------------------------------------------------------
// out[c][r] = in[r][c] ; out has Rpad >= R columns, columns R..Rpad-1 are zero-filled (GEMM K must be a multiple of 16)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                        int ldin, int Rpad) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < C) ? in[(long long)r * ldin + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < Rpad) out[(long long)c * Rpad + r] = tile[tx][i];
  }
}

// out[n] = sum_m x[m][n]  (fixed order: each block owns 64 columns, 4 row-lanes, then a 4-way tree)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int ld,
                                                     int accumulate) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float s = 0.f;
  if (n < N)
    for (int m = rl; m < M; m += 4) s += x[(long long)m * ld + n];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && n < N) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    out[n] = accumulate ? out[n] + t : t;
  }
}

// two-stage, fixed-order column sum for tall inputs: block (cb, rs) sums the rows of slice rs for 64 columns ->
// part[rs][n]; colsum_finish adds the RS partials in slice order (deterministic).  Grid (ceil(N/64), RS).
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, float* __restrict__ part, int M, int N, int ld,
                                                          int rows_per) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  float s = 0.f;
  if (n < N)
    for (int m = m0 + rl; m < m1; m += 4) s += x[(long long)m * ld + n];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && n < N)
    part[(long long)blockIdx.y * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int N, int RS, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float t = 0.f;
  for (int r = 0; r < RS; ++r) t += part[(long long)r * N + n];
  out[n] = accumulate ? out[n] + t : t;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = gelu_f(x[i]);
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = dy[i] * gelu_grad(x[i]);
}

// LayerNorm backward, row part (one workgroup per row):
//   xhat = (x-mean)*rstd ; g = dy*gamma ; dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+ dres)
// also stores mean/rstd per row for the column reductions of dgamma/dbeta.
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ dres,
                                                          float* __restrict__ dx, float* __restrict__ stats /*(M,2)*/, int D) {
  __shared__ float red[12];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (long long)m * D;
  const float* dyr = dy + (long long)m * D;
  float s = 0.f;
  for (int c = tid; c < D; c += 256) s += xr[c];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D;
  float q = 0.f;
  for (int c = tid; c < D; c += 256) { const float d = xr[c] - mean; q += d * d; }
  q = wave_sum(q);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rstd = rsqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)D + 1e-5f);
  float a = 0.f, b = 0.f;
  for (int c = tid; c < D; c += 256) {
    const float g = dyr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    a += g; b += g * xh;
  }
  a = wave_sum(a); b = wave_sum(b);
  __syncthreads();
  if (lane == 0) { red[wave] = a; red[4 + wave] = b; }
  __syncthreads();
  const float ma = ((red[0] + red[1]) + (red[2] + red[3])) / (float)D, mb = ((red[4] + red[5]) + (red[6] + red[7])) / (float)D;
  for (int c = tid; c < D; c += 256) {
    const float g = dyr[c] * gamma[c], xh = (xr[c] - mean) * rstd;
    float v = rstd * (g - ma - xh * mb);
    if (dres) v += dres[(long long)m * D + c];
    dx[(long long)m * D + c] = v;
  }
  if (tid == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
}

// The same row part with ONE WAVE PER ROW (round 5): the row lives in registers (NE = D / 64 elements per lane, element lane + 64 i:
// every load / store is a full 256-byte line per wave), read once, every reduction is a wavefront butterfly - no LDS, no barriers,
// four rows per workgroup.  The block-per-row form above re-reads the row four times through three barriers and was launch-latency
// bound (8.4 us for 499 rows).  Optional second output: dx2 = nn.Dropout(dx) with the counter-hash mask of `drop_seed` - the gradient the
// next GEMMs of the backward pass need when the forward dropped this tensor (mingpt.py:90,105), instead of a separate dropout launch.
template <int NE>
__global__ __launch_bounds__(256) void ln_bwd_rows_wave_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ dres,
                                                               float* __restrict__ dx, float* __restrict__ stats, float* __restrict__ dx2,
                                                               int M, float drop_p, unsigned drop_seed, const unsigned* __restrict__ drop_seed_dev) {
  constexpr int D = 64 * NE;
  if (drop_seed_dev) drop_seed = *drop_seed_dev;      // the seed lives in device memory (a captured hipGraph of the training step is reused across steps)
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  const long long rb = (long long)m * D;
  float xv[NE], gv[NE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) { xv[i] = x[rb + lane + 64 * i]; s += xv[i]; }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) { xv[i] -= mean; q += xv[i] * xv[i]; }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    xv[i] *= rstd;                                              // xhat
    gv[i] = dy[rb + lane + 64 * i] * gamma[lane + 64 * i];
    a += gv[i]; b += gv[i] * xv[i];
  }
  const float ma = wave_sum(a) / (float)D, mb = wave_sum(b) / (float)D;
  const float inv_keep = 1.0f / (1.0f - drop_p);
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const long long o = rb + lane + 64 * i;
    float v = rstd * (gv[i] - ma - xv[i] * mb);
    if (dres) v += dres[o];
    dx[o] = v;
    if (dx2) dx2[o] = v * sfmi_dropout_mul(drop_seed, (unsigned)o, drop_p, inv_keep);
  }
  if (lane == 0) { stats[2 * m] = mean; stats[2 * m + 1] = rstd; }
}

// LayerNorm backward, parameter part: dgamma[c] += sum_m dy*xhat ; dbeta[c] += sum_m dy
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int M, int D) {
  __shared__ float rg[4][64], rb[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  float g = 0.f, b = 0.f;
  if (c < D)
    for (int m = rl; m < M; m += 4) {
      const float d = dy[(long long)m * D + c];
      g += d * (x[(long long)m * D + c] - stats[2 * m]) * stats[2 * m + 1];
      b += d;
    }
  rg[rl][threadIdx.x & 63] = g; rb[rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && c < D) {
    dgamma[c] += (rg[0][threadIdx.x] + rg[1][threadIdx.x]) + (rg[2][threadIdx.x] + rg[3][threadIdx.x]);
    dbeta[c] += (rb[0][threadIdx.x] + rb[1][threadIdx.x]) + (rb[2][threadIdx.x] + rb[3][threadIdx.x]);
  }
}

// the same parameter sums in two stages for tall inputs (grid (ceil(D/64), RS)): part[rs][0][c] = dgamma, part[rs][1][c] = dbeta
__global__ __launch_bounds__(256) void ln_bwd_params_part_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ stats, float* __restrict__ part, int M, int D,
                                                                 int rows_per) {
  __shared__ float rg[4][64], rb[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  float g = 0.f, b = 0.f;
  if (c < D)
    for (int m = m0 + rl; m < m1; m += 4) {
      const float d = dy[(long long)m * D + c];
      g += d * (x[(long long)m * D + c] - stats[2 * m]) * stats[2 * m + 1];
      b += d;
    }
  rg[rl][threadIdx.x & 63] = g; rb[rl][threadIdx.x & 63] = b;
  __syncthreads();
  if (rl == 0 && c < D) {
    float* o = part + (long long)blockIdx.y * 2 * D;
    o[c] = (rg[0][threadIdx.x] + rg[1][threadIdx.x]) + (rg[2][threadIdx.x] + rg[3][threadIdx.x]);
    o[D + c] = (rb[0][threadIdx.x] + rb[1][threadIdx.x]) + (rb[2][threadIdx.x] + rb[3][threadIdx.x]);
  }
}
__global__ void ln_bwd_params_finish_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, int D,
                                            int RS) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float g = 0.f, b = 0.f;
  for (int r = 0; r < RS; ++r) { g += part[(long long)r * 2 * D + c]; b += part[(long long)r * 2 * D + D + c]; }
  dgamma[c] += g; dbeta[c] += b;
}


// ------------------------------------------------------------------------------------------------
// All column reductions of ONE transformer block's backward in ONE launch (round 5): the four bias gradients (column sums of the
// four dY's, mingpt.py:46-111 Linear biases) and the two LayerNorm parameter gradients (dgamma = sum_m dy * xhat, dbeta = sum_m dy).
// They are only consumed by the optimizer, so they are deferred to the end of the block's backward and share a launch (round 4:
// 8 colsum + 4 LayerNorm-parameter launches per block).  A job is a (kind, M x N operand(s), outputs) record; the grid is
// (sum of the jobs' 64-column blocks, RS row slices).  RS = 1: direct.  RS > 1 (tall inputs): slice partials go out as agent-scope
// relaxed atomics (write-through), the block that draws the last ticket of its column block adds them in slice order
// (deterministic) - the wave-level protocol of csrc/gpt.hip's split-K, here per workgroup (drain, barrier, ticket).
// Thread layout: 16 row lanes x 16 float4 column quads; fixed-order tree over the row lanes.
// ------------------------------------------------------------------------------------------------
constexpr int CR_MAX_JOBS = 8;
struct ColJob {
  const float* a;       // (M, N; ld): the summed operand (dY)
  const float* x;       // kind 1: the LayerNorm input (M, N; ld)
  const float* stats;   // kind 1: (M, 2) row mean / rstd (ln_bwd_rows_kernel)
  float* out;           // kind 0: column sums; kind 1: dgamma
  float* out2;          // kind 1: dbeta
  int N, ld, kind, blk0;   // blk0: first column block of this job in grid.x
};
struct ColJobs { ColJob j[CR_MAX_JOBS]; int njobs, M, accumulate, RS; float* part; int* cnt; };

__global__ __launch_bounds__(256) void col_reduce_kernel(ColJobs J) {
  __shared__ float red[2][16][64];
  __shared__ int last_flag;
  int ji = 0;
#pragma unroll
  for (int t = 1; t < CR_MAX_JOBS; ++t)
    if (t < J.njobs && (int)blockIdx.x >= J.j[t].blk0) ji = t;
  const ColJob& jb = J.j[ji];
  const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
  const int n = ((int)blockIdx.x - jb.blk0) * 64 + 4 * cq;
  const int rows_per = (J.M + J.RS - 1) / J.RS;
  const int m0 = blockIdx.y * rows_per, m1 = min(J.M, m0 + rows_per);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
  if (n < jb.N) {
    if (jb.kind == 0) {
      for (int m = m0 + rl; m < m1; m += 16) s = s + *reinterpret_cast<const f32x4*>(jb.a + (long long)m * jb.ld + n);
    } else {
      for (int m = m0 + rl; m < m1; m += 16) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(jb.a + (long long)m * jb.ld + n);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(jb.x + (long long)m * jb.ld + n);
        const float mean = jb.stats[2 * m], rstd = jb.stats[2 * m + 1];
        s = s + d;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += d[e] * (xv[e] - mean) * rstd;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][rl][4 * cq + e] = s[e]; red[1][rl][4 * cq + e] = g[e]; }
  __syncthreads();
  // threads 0..63: column sums; 64..127: the dgamma sums (kind 1)
  const int which = tid >> 6, c = tid & 63;
  float t = 0.f;
  if (tid < 128) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[which][r][c];
  }
  const int nc = ((int)blockIdx.x - jb.blk0) * 64 + c;
  if (J.RS > 1) {
    float* part = J.part + ((long long)blockIdx.x * J.RS) * 128;
    if (tid < 128) __hip_atomic_store(part + (long long)blockIdx.y * 128 + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(J.cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_flag = ticket == J.RS - 1;
      if (last_flag) __hip_atomic_store(J.cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_flag) return;
    t = 0.f;
    if (tid < 128)
      for (int r = 0; r < J.RS; ++r) t += __hip_atomic_load(part + (long long)r * 128 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < 128 && nc < jb.N) {
    float* o = which == 0 ? (jb.kind == 0 ? jb.out : jb.out2) : (jb.kind == 1 ? jb.out : nullptr);
    if (o) o[nc] = J.accumulate ? o[nc] + t : t;
  }
}

// ------------------------------------------------------------------------------------------------
// Attention backward on the matrix cores (16x16x4 f32 MFMA): dqkv (B*L,3D) from qkv, y, dy (mingpt.py:73-91), head dim 64.
// Shared conventions (csrc/gpt.hip attn_prefill_mfma_kernel): 4 waves, wave w owns 16 rows of the block; an A operand is
// (row = lane & 15, k = 4 kk + (lane >> 4)), a B operand (k = 4 kk + (lane >> 4), col = lane & 15), C/D register j holds
// row 4 (lane >> 4) + j, column lane & 15.  LDS tiles are 64 rows x stride 68: "row-major" operand reads (16 rows x 4
// k-columns) are conflict-free, "k-major" reads (4 rows x 16 columns) are ~2-way.
//   stats : lse[row] = logsumexp_k(scale q.k) (64 MFMAs per 16x64 tile), delta[row] = sum_d dO O
//   dq    : per query block:  S = Q K^T, dP = dO V^T, dS = exp(S - lse) (dP - delta), dQ += dS K          (192 MFMAs / tile)
//   dkv   : per key block:    S^T = K Q^T, dP^T = V dO^T, P^T, dS^T; dV += P^T dO, dK += dS^T Q           (256 MFMAs / tile)
// ------------------------------------------------------------------------------------------------
constexpr int AB_S = 68;
#define AB_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x4f32((a_), (b_), (c_), 0, 0, 0)

// stage rows [r0, r0+64) of one (b, h) 64-wide slice of `src` (row stride ld floats) into a stride-68 LDS tile
__device__ __forceinline__ void ab_stage(float* __restrict__ dst, const float* __restrict__ src, long long row_base, int r0, int L, int ld,
                                         int tid) {
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    const int t = min(r0 + r, L - 1);
    *reinterpret_cast<f32x4*>(&dst[r * AB_S + 4 * c]) = *reinterpret_cast<const f32x4*>(src + (row_base + t) * ld + 4 * c);
  }
}

__global__ __launch_bounds__(256) void attn_stats_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ y,
                                                              const float* __restrict__ dy, float* __restrict__ lse,
                                                              float* __restrict__ delta, int L, int D, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * AB_S];
  const int b = blockIdx.x, h = blockIdx.y, qb = blockIdx.z, H = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4, q0 = qb * 64;
  const long long rb = (long long)b * L;
  const int trow = min(q0 + 16 * wave + lr, L - 1);
  float qf[16], dl = 0.f;
  {
    const float* qp = qkv + (rb + trow) * 3 * D + h * 64 + lq;
    const float* yp = y + (rb + trow) * D + h * 64 + lq;
    const float* dp = dy + (rb + trow) * D + h * 64 + lq;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { qf[kk] = qp[4 * kk] * scale; dl += yp[4 * kk] * dp[4 * kk]; }
  }
  dl += __shfl_xor(dl, 16, 64); dl += __shfl_xor(dl, 32, 64);      // all four lane groups now hold delta of row lr
  float mrun[4], lrun[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.f; }
  const int kend = min(L, q0 + 64);
  for (int k0 = 0; k0 < kend; k0 += 64) {
    __syncthreads();
    ab_stage(Ks, qkv + D + h * 64, rb, k0, L, 3 * D, tid);
    __syncthreads();
    f32x4 sacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kp = &Ks[(16 * t + lr) * AB_S + lq];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) sacc[t] = AB_MFMA(qf[kk], kp[4 * kk], sacc[t]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qrow = q0 + 16 * wave + 4 * lq + j;
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int key = k0 + 16 * t + lr;
        if (key > qrow || key >= L) sacc[t][j] = -INFINITY;
        mx = fmaxf(mx, sacc[t][j]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
      const float mnew = fmaxf(mrun[j], mx), ms = mnew == -INFINITY ? 0.f : mnew;
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) ps += __expf(sacc[t][j] - ms);
      ps += __shfl_xor(ps, 1, 64); ps += __shfl_xor(ps, 2, 64); ps += __shfl_xor(ps, 4, 64); ps += __shfl_xor(ps, 8, 64);
      lrun[j] = lrun[j] * __expf(mrun[j] - ms) + ps;
      mrun[j] = mnew;
    }
  }
  const long long sb = ((long long)b * H + h) * L;
  if (lr == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tq = q0 + 16 * wave + 4 * lq + j;
      if (tq < L) lse[sb + tq] = mrun[j] + __logf(lrun[j]);
    }
  }
  if (lq == 0 && q0 + 16 * wave + lr < L) delta[sb + q0 + 16 * wave + lr] = dl;
}

// delta[b][h][t] = sum_d dO[b,t,h,d] * O[b,t,h,d]: one wave per row, lane l covers the H floats l*H .. l*H+H-1 of the row (D = 64 H, so a
// head is 64 / H consecutive lanes).  Needed by both halves of the backward; a few microseconds.
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ delta,
                                                         int B, int L, int D, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * L) return;
  const float* yp = y + (long long)row * D + lane * H;
  const float* dp = dy + (long long)row * D + lane * H;
  float sacc = 0.f;
  for (int e = 0; e < H; ++e) sacc += yp[e] * dp[e];
  const int lph = 64 / H;      // lanes per head
  for (int o = 1; o < lph; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
  if (lane % lph == 0) {
    const int b = row / L, t = row - b * L, h = lane / lph;
    delta[((long long)b * H + h) * L + t] = sacc;
  }
}

// one 64-row tile of a (b, h) slice: global -> registers (4 float4 per thread) -> stride-68 LDS tile.  The loads of block i+1 are
// issued right after block i has been handed to LDS, so their latency runs under the block's 192 / 256 MFMAs per wave.
template <int GT>      // GT threads of a wave group move one 64 x 64 tile: 1024 / GT float4 each
__device__ __forceinline__ void ab_load(f32x4 (&r)[1024 / GT], const float* __restrict__ src, long long row_base, int r0, int L, int ld, int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / GT; ++it) {
    const int i = tid + GT * it, rr = i >> 4, c = i & 15;
    const int t = min(r0 + rr, L - 1);
    r[it] = *reinterpret_cast<const f32x4*>(src + (row_base + t) * ld + 4 * c);
  }
}
template <int GT>
__device__ __forceinline__ void ab_store(float* __restrict__ dst, const f32x4 (&r)[1024 / GT], int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / GT; ++it) {
    const int i = tid + GT * it, rr = i >> 4, c = i & 15;
    *reinterpret_cast<f32x4*>(&dst[rr * AB_S + 4 * c]) = r[it];
  }
}

// dQ of query block qb (the former attn_bwd_dq_mfma_kernel: same arithmetic, same order).
// Tiling: a workgroup owns 16 * RW query rows (RW row waves) and NG wave groups; group g walks the key blocks g, g + NG, ... with its own
// LDS tiles and the partial dQ's are added through LDS at the end.  <RW 4, NG 1> is the form of rounds 1-4 (bit-identical).  Small
// launches (batch 1: everything resident at once, the launch lasts as long as its heaviest WAVE - the last query rows walk every key
// block) use <RW 2, NG 2>: the same rows' work spread over twice the SIMDs, twice the workgroups to balance the causal triangle.
template <int RW, int NG>
__device__ __forceinline__ void attn_bwd_dq_role(float* __restrict__ smem_all, const float* __restrict__ qkv, const float* __restrict__ dy,
                                                 const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                 int b, int h, int H, int qb, int L, int D, float scale, float drop_p, unsigned drop_seed) {
  constexpr int GT = 64 * RW, GROUP_FLOATS = 2 * 64 * AB_S + 2 * RW * 16 * AB_S;
  const int grp = NG == 1 ? 0 : (int)(threadIdx.x / GT);
  float* smem = smem_all + grp * GROUP_FLOATS;
  float* Ks = smem;
  float* Vs = smem + 64 * AB_S;
  const int tid = threadIdx.x % GT, lane = tid & 63, wave = tid >> 6;
  float* Tw = smem + 2 * 64 * AB_S + wave * 16 * AB_S;
  const int lr = lane & 15, lq = lane >> 4, q0 = qb * 16 * RW;
  const long long rb = (long long)b * L, sb = ((long long)b * H + h) * L;
  const int trow = min(q0 + 16 * wave + lr, L - 1);
  const int kend = min(L, q0 + 16 * RW), nblk = (kend + 63) / 64, rounds = (nblk + NG - 1) / NG;
  f32x4 pk[1024 / GT], pv[1024 / GT];
  if (grp < nblk) {
    ab_load<GT>(pk, qkv + D + h * 64, rb, 64 * grp, L, 3 * D, tid);
    ab_load<GT>(pv, qkv + 2 * D + h * 64, rb, 64 * grp, L, 3 * D, tid);
  }
  float qf[16], dof[16];
  {
    const float* qp = qkv + (rb + trow) * 3 * D + h * 64 + lq;
    const float* dp = dy + (rb + trow) * D + h * 64 + lq;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { qf[kk] = qp[4 * kk] * scale; dof[kk] = dp[4 * kk]; }
  }
  float ls[4], dl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tq = min(q0 + 16 * wave + 4 * lq + j, L - 1);
    ls[j] = lse[sb + tq]; dl[j] = delta[sb + tq];
  }
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < rounds; ++r) {
    const int k0 = 64 * (r * NG + grp);
    const bool active = k0 < kend;           // group-uniform; the barriers below are executed by every wave of the workgroup
    __syncthreads();
    if (active) {
      ab_store<GT>(Ks, pk, tid);
      ab_store<GT>(Vs, pv, tid);
    }
    __syncthreads();
    if (!active) continue;
    if (k0 + 64 * NG < kend) {
      ab_load<GT>(pk, qkv + D + h * 64, rb, k0 + 64 * NG, L, 3 * D, tid);
      ab_load<GT>(pv, qkv + 2 * D + h * 64, rb, k0 + 64 * NG, L, 3 * D, tid);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
      const float* kp = &Ks[(16 * t + lr) * AB_S + lq];
      const float* vp = &Vs[(16 * t + lr) * AB_S + lq];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { sa = AB_MFMA(qf[kk], kp[4 * kk], sa); pa = AB_MFMA(dof[kk], vp[4 * kk], pa); }
      const int key = k0 + 16 * t + lr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qrow = q0 + 16 * wave + 4 * lq + j;
        // attention dropout (mingpt.py:85): O = (P o M) V  =>  dP = (dO V^T) o M ; delta = rowsum(dO o O) is unchanged
        const float mk = drop_p > 0.f ? sfmi_dropout_mul(drop_seed, (unsigned)(((b * H + h) * L + qrow) * L + key), drop_p, 1.0f / (1.0f - drop_p)) : 1.0f;
        const float ds = (key > qrow || key >= L) ? 0.f : __expf(sa[j] - ls[j]) * (pa[j] * mk - dl[j]);
        Tw[(4 * lq + j) * AB_S + 16 * t + lr] = ds;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const float* tp = &Tw[lr * AB_S + lq];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float a = tp[4 * kk];
      const float* kb = &Ks[(4 * kk + lq) * AB_S + lr];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = AB_MFMA(a, kb[16 * dt], dq[dt]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (NG > 1) {      // group 1's partial sums -> LDS (its own tiles are free after the barrier) -> group 0 adds them, in group order
    __syncthreads();
    static_assert(NG <= 2, "the exchange below adds ONE other group");
    float* xch = smem_all + GROUP_FLOATS;        // GT threads x 16 floats
    if (grp == 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(&xch[(dt * GT + tid) * 4]) = dq[dt];
    }
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = dq[dt] + *reinterpret_cast<const f32x4*>(&xch[(dt * GT + tid) * 4]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tq = q0 + 16 * wave + 4 * lq + j;
    if (tq >= L) continue;
    float* o = dqkv + (rb + tq) * 3 * D + h * 64 + lr;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[16 * dt] = dq[dt][j] * scale;
  }
}

// dK / dV of key block kb_ (the former attn_bwd_dkv_mfma_kernel: same arithmetic, same order); NG wave groups as above, over the query blocks
template <int RW, int NG>
__device__ __forceinline__ void attn_bwd_dkv_role(float* __restrict__ smem_all, const float* __restrict__ qkv, const float* __restrict__ dy,
                                                  const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                  int b, int h, int H, int kb_, int L, int D, float scale, float drop_p, unsigned drop_seed) {
  constexpr int GT = 64 * RW, GROUP_FLOATS = 2 * 64 * AB_S + 2 * RW * 16 * AB_S;
  const int grp = NG == 1 ? 0 : (int)(threadIdx.x / GT);
  float* smem = smem_all + grp * GROUP_FLOATS;
  float* Qs = smem;
  float* Os = smem + 64 * AB_S;
  const int tid = threadIdx.x % GT, lane = tid & 63, wave = tid >> 6;
  float* Pw = smem + 2 * 64 * AB_S + wave * 16 * AB_S;
  float* Sw = smem + 2 * 64 * AB_S + RW * 16 * AB_S + wave * 16 * AB_S;
  const int lr = lane & 15, lq = lane >> 4, k0 = kb_ * 16 * RW;
  const int qs = (k0 >> 6) << 6;      // the 64-aligned query block that holds the first query row >= k0
  const long long rb = (long long)b * L, sb = ((long long)b * H + h) * L;
  const int krow = min(k0 + 16 * wave + lr, L - 1);
  const int nblk = (L - qs + 63) / 64, rounds = (nblk + NG - 1) / NG;
  f32x4 pq[1024 / GT], po[1024 / GT];
  if (grp < nblk) {
    ab_load<GT>(pq, qkv + h * 64, rb, qs + 64 * grp, L, 3 * D, tid);
    ab_load<GT>(po, dy + h * 64, rb, qs + 64 * grp, L, D, tid);
  }
  float kf[16], vf[16];
  {
    const float* kp = qkv + (rb + krow) * 3 * D + D + h * 64 + lq;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) { kf[kk] = kp[4 * kk] * scale; vf[kk] = kp[D + 4 * kk]; }
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; }
  for (int r = 0; r < rounds; ++r) {
    const int q0 = qs + 64 * (r * NG + grp);
    const bool active = q0 < L;
    __syncthreads();
    if (active) {
      ab_store<GT>(Qs, pq, tid);
      ab_store<GT>(Os, po, tid);
    }
    __syncthreads();
    if (!active) continue;
    if (q0 + 64 * NG < L) {
      ab_load<GT>(pq, qkv + h * 64, rb, q0 + 64 * NG, L, 3 * D, tid);
      ab_load<GT>(po, dy + h * 64, rb, q0 + 64 * NG, L, D, tid);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, pa = {0.f, 0.f, 0.f, 0.f};
      const float* qp = &Qs[(16 * t + lr) * AB_S + lq];
      const float* op = &Os[(16 * t + lr) * AB_S + lq];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) { sa = AB_MFMA(kf[kk], qp[4 * kk], sa); pa = AB_MFMA(vf[kk], op[4 * kk], pa); }
      const int qcol = q0 + 16 * t + lr;                       // this lane's column = a query row
      const int qc = min(qcol, L - 1);
      const float lsq = lse[sb + qc], dlq = delta[sb + qc];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = k0 + 16 * wave + 4 * lq + j;
        const bool live = qcol < L && key <= qcol && key < L;
        const float pv = live ? __expf(sa[j] - lsq) : 0.f;
        const float mk = drop_p > 0.f ? sfmi_dropout_mul(drop_seed, (unsigned)(((b * H + h) * L + qc) * L + key), drop_p, 1.0f / (1.0f - drop_p)) : 1.0f;
        Pw[(4 * lq + j) * AB_S + 16 * t + lr] = pv * mk;                 // dV += (P o M)^T dO
        Sw[(4 * lq + j) * AB_S + 16 * t + lr] = pv * (pa[j] * mk - dlq);  // dS^T = P^T o (dP^T o M - delta)
      }
    }
    __builtin_amdgcn_wave_barrier();
    const float* pp = &Pw[lr * AB_S + lq];
    const float* sp = &Sw[lr * AB_S + lq];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float pa = pp[4 * kk], sa = sp[4 * kk];
      const float* ob = &Os[(4 * kk + lq) * AB_S + lr];
      const float* qb2 = &Qs[(4 * kk + lq) * AB_S + lr];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dv[dt] = AB_MFMA(pa, ob[16 * dt], dv[dt]); dk[dt] = AB_MFMA(sa, qb2[16 * dt], dk[dt]); }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (NG > 1) {
    __syncthreads();
    static_assert(NG <= 2, "the exchange below adds ONE other group");
    float* xch = smem_all + GROUP_FLOATS;        // GT threads x 32 floats
    if (grp == 1) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<f32x4*>(&xch[(dt * GT + tid) * 4]) = dk[dt];
        *reinterpret_cast<f32x4*>(&xch[((4 + dt) * GT + tid) * 4]) = dv[dt];
      }
    }
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[dt] = dk[dt] + *reinterpret_cast<const f32x4*>(&xch[(dt * GT + tid) * 4]);
      dv[dt] = dv[dt] + *reinterpret_cast<const f32x4*>(&xch[((4 + dt) * GT + tid) * 4]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tk = k0 + 16 * wave + 4 * lq + j;
    if (tk >= L) continue;
    float* o = dqkv + (rb + tk) * 3 * D + D + h * 64 + lr;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { o[16 * dt] = dk[dt][j] * scale; o[D + 16 * dt] = dv[dt][j]; }
  }
}

// Both halves of the attention backward in ONE launch (round 5; rounds 1-4: stats + dq + dkv = three launches, the first of which
// recomputed Q K^T only to get the row log-sum-exps the forward pass already had): grid (B, H, 2 * ceil(L / 64)); z even: dQ of query
// block nqb - 1 - z / 2, z odd: dK / dV of key block z / 2 - the longest loops of either kind are dispatched first.  The two kinds are
// independent given lse and delta, so the launch has twice the workgroups of either (256 at batch 1: every CU gets one), and the
// causal imbalance of one kind (1 .. nqb blocks per workgroup) is filled by the other.
// ------------------------------------------------------------------------------------------------
// Training forward of the causal attention for SMALL launches (round 5; batch 1 = 16 heads x 8 query blocks = 128 workgroups of the
// prefill kernel for 256 CUs, and the last query block walks 8 key blocks one after the other with one wave per SIMD: 43 us per layer
// for 0.26 GFLOP).  A workgroup owns 16 * RW query rows; NG wave groups walk the key blocks g, g + NG, ... with their own K / V tiles and
// their own online-softmax state (m, l, O); the groups' states are merged through LDS at the end:
//   m = max_g m_g ;  l = sum_g l_g exp(m_g - m) ;  O = sum_g O_g exp(m_g - m) ;  y = O / l ;  lse = m + log l
// Same MFMA shapes and LDS conventions as the backward kernels below (16x16x4 f32, stride-68 tiles, C/D register j = row 4 (lane>>4) + j).
// Attention dropout as in csrc/gpt.hip:attn_prefill_mfma_kernel (mask of element (b, h, query, key), applied to P after the row sum).
// ------------------------------------------------------------------------------------------------
template <int RW, int NG>
__global__ __launch_bounds__(64 * RW * NG) void attn_train_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ y,
                                                                      float* __restrict__ lse, int L, int D, float scale, float drop_p,
                                                                      unsigned drop_seed, const unsigned* __restrict__ drop_seed_dev) {
  constexpr int GT = 64 * RW, NIT = 1024 / GT, GROUP_FLOATS = 2 * 64 * AB_S + RW * 16 * AB_S;
  if (drop_seed_dev) drop_seed = *drop_seed_dev;
  __shared__ __attribute__((aligned(16))) float smem_all[NG * GROUP_FLOATS];
  const int b = blockIdx.x, h = blockIdx.y, H = gridDim.y, qb = blockIdx.z;
  const int grp = NG == 1 ? 0 : (int)(threadIdx.x / GT), tid = threadIdx.x % GT, lane = tid & 63, wave = tid >> 6;
  float* Ks = smem_all + grp * GROUP_FLOATS;
  float* Vs = Ks + 64 * AB_S;
  float* Pw = Vs + 64 * AB_S + wave * 16 * AB_S;
  const int lr = lane & 15, lq = lane >> 4, q0 = qb * 16 * RW;
  const long long rb = (long long)b * L;
  float qf[16];
  {
    const int tq = min(q0 + 16 * wave + lr, L - 1);
    const float* qp = qkv + (rb + tq) * 3 * D + h * 64 + lq;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) qf[kk] = qp[4 * kk] * scale;
  }
  float mrun[4], lrun[4];
  f32x4 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { mrun[j] = -INFINITY; lrun[j] = 0.f; }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kend = min(L, q0 + 16 * RW), nblk = (kend + 63) / 64, rounds = (nblk + NG - 1) / NG;
  f32x4 pk[NIT], pv[NIT];
  if (grp < nblk) {
    ab_load<GT>(pk, qkv + D + h * 64, rb, 64 * grp, L, 3 * D, tid);
    ab_load<GT>(pv, qkv + 2 * D + h * 64, rb, 64 * grp, L, 3 * D, tid);
  }
  for (int r = 0; r < rounds; ++r) {
    const int k0 = 64 * (r * NG + grp);
    const bool active = k0 < kend;            // group-uniform; every wave of the workgroup executes the two barriers
    __syncthreads();
    if (active) {
      ab_store<GT>(Ks, pk, tid);
      ab_store<GT>(Vs, pv, tid);
    }
    __syncthreads();
    if (!active) continue;
    if (k0 + 64 * NG < kend) {
      ab_load<GT>(pk, qkv + D + h * 64, rb, k0 + 64 * NG, L, 3 * D, tid);
      ab_load<GT>(pv, qkv + 2 * D + h * 64, rb, k0 + 64 * NG, L, 3 * D, tid);
    }
    // (tiles of the diagonal block that lie wholly above the wave's rows are fully masked - p == 0 exactly - and are computed anyway:
    //  at most three wasted tiles per wave and query tile)
    f32x4 sacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      sacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kp = &Ks[(16 * t + lr) * AB_S + lq];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) sacc[t] = AB_MFMA(qf[kk], kp[4 * kk], sacc[t]);
    }
    float corr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int qrow = q0 + 16 * wave + 4 * lq + j;
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int key = k0 + 16 * t + lr;
        if (key > qrow || key >= L) sacc[t][j] = -INFINITY;
        mx = fmaxf(mx, sacc[t][j]);
      }
      mx = row16_max(mx);
      const float mnew = fmaxf(mrun[j], mx), ms = mnew == -INFINITY ? 0.f : mnew;
      corr[j] = __expf(mrun[j] - ms);
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float pe = __expf(sacc[t][j] - ms);
        ps += pe;                        // the softmax denominator is the undropped sum (mingpt.py:84-85)
        if (drop_p > 0.f)
          pe *= sfmi_dropout_mul(drop_seed, (unsigned)(((b * H + h) * L + qrow) * L + k0 + 16 * t + lr), drop_p, 1.0f / (1.0f - drop_p));
        Pw[(4 * lq + j) * AB_S + 16 * t + lr] = pe;
      }
      ps = row16_sum(ps);
      lrun[j] = lrun[j] * corr[j] + ps;
      mrun[j] = mnew;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[dt][j] *= corr[j];
    __builtin_amdgcn_wave_barrier();
    const float* pp = &Pw[lr * AB_S + lq];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {      // keys 4 kk + lq
      const float pa = pp[4 * kk];
      const float* vb = &Vs[(4 * kk + lq) * AB_S + lr];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = AB_MFMA(pa, vb[16 * dt], o[dt]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (NG > 1) {      // merge the groups' (m, l, O) states: group g > 0 publishes, group 0 folds them in, in group order
    __syncthreads();
    float* xch = smem_all + GROUP_FLOATS;        // (NG - 1) x 24 x GT floats: inside the other groups' (now free) tiles
    if (grp > 0) {
      float* x = xch + (grp - 1) * 24 * GT;
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j * GT + tid] = mrun[j]; x[(4 + j) * GT + tid] = lrun[j]; }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[(8 + 4 * dt + j) * GT + tid] = o[dt][j];
    }
    __syncthreads();
    if (grp != 0) return;
    for (int g = 1; g < NG; ++g) {
      const float* x = xch + (g - 1) * 24 * GT;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float mg = x[j * GT + tid], lg = x[(4 + j) * GT + tid];
        const float mnew = fmaxf(mrun[j], mg), ms = mnew == -INFINITY ? 0.f : mnew;
        const float ca = __expf(mrun[j] - ms), cb = __expf(mg - ms);      // exp(-inf) = 0: an empty state contributes nothing
        lrun[j] = lrun[j] * ca + lg * cb;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt][j] = o[dt][j] * ca + x[(8 + 4 * dt + j) * GT + tid] * cb;
        mrun[j] = mnew;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tq = q0 + 16 * wave + 4 * lq + j;
    if (tq >= L) continue;
    const float inv = 1.0f / lrun[j];
    float* yp = y + (rb + tq) * D + h * 64 + lr;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) yp[16 * dt] = o[dt][j] * inv;
    if (lr == 0) lse[((long long)b * H + h) * L + tq] = mrun[j] + __logf(lrun[j]);
  }
}

template <int RW, int NG>
__global__ __launch_bounds__(64 * RW * NG) void attn_bwd_fused_kernel(const float* __restrict__ qkv, const float* __restrict__ dy,
                                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                                      float* __restrict__ dqkv, int L, int D, float scale, float drop_p,
                                                                      unsigned drop_seed, const unsigned* __restrict__ drop_seed_dev) {
  __shared__ __attribute__((aligned(16))) float smem[NG * (2 * 64 * AB_S + 2 * RW * 16 * AB_S)];
  if (drop_seed_dev) drop_seed = *drop_seed_dev;
  const int b = blockIdx.x, h = blockIdx.y, H = gridDim.y, z = blockIdx.z, nqb = gridDim.z >> 1;
  if (z & 1) attn_bwd_dkv_role<RW, NG>(smem, qkv, dy, lse, delta, dqkv, b, h, H, z >> 1, L, D, scale, drop_p, drop_seed);
  else attn_bwd_dq_role<RW, NG>(smem, qkv, dy, lse, delta, dqkv, b, h, H, nqb - 1 - (z >> 1), L, D, scale, drop_p, drop_seed);
}

// small launches (every workgroup resident at once: the heaviest wave is the launch's duration) run 32-row tiles x two wave groups
static void attn_bwd_fused_launch(hipStream_t st, const float* qkv, const float* dy, const float* lse, const float* delta, float* dqkv, int B,
                                  int L, int D, int H, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev) {
  const int nqb = (L + 63) / 64;
  if ((long long)B * H * 2 * nqb <= 256 && nqb > 1) {
    const int n32 = (L + 31) / 32;
    hipLaunchKernelGGL((attn_bwd_fused_kernel<2, 2>), dim3(B, H, 2 * n32), dim3(256), 0, st, qkv, dy, lse, delta, dqkv, L, D, 0.125f, drop_p, drop_seed, drop_seed_dev);
  } else {
    hipLaunchKernelGGL((attn_bwd_fused_kernel<4, 1>), dim3(B, H, 2 * nqb), dim3(256), 0, st, qkv, dy, lse, delta, dqkv, L, D, 0.125f, drop_p, drop_seed, drop_seed_dev);
  }
}

// softmax cross-entropy: loss_row[m] = lse - logit[target] ; dlogits = (softmax - onehot) * scale for rows with
// t >= t0 (row m = (b, t), t = m % L), zero otherwise (F.cross_entropy mean over B*L_z rows, shapeformer.py:134-139)
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ target,
                                                         float* __restrict__ loss_rows, float* __restrict__ dlogits, int V, int ld,
                                                         int L, int t0, float scale) {
  __shared__ float red[8];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (long long)m * ld;
  float* drow = dlogits + (long long)m * ld;
  const bool active = (m % L) >= t0;
  if (!active) {
    for (int v = tid; v < ld; v += 256) drow[v] = 0.f;
    if (tid == 0) loss_rows[m] = 0.f;
    return;
  }
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
  const float tot = (red[4] + red[5]) + (red[6] + red[7]);
  const int tg = target[m];
  for (int v = tid; v < ld; v += 256) {
    float g = 0.f;
    if (v < V) g = (__expf(row[v] - mx) / tot - (v == tg ? 1.f : 0.f)) * scale;
    drow[v] = g;
  }
  if (tid == 0) loss_rows[m] = mx + __logf(tot) - row[tg];
}




// ---------------------------------------------------------------------------------------------------------------
// embedding gradients: acc[idx[m]][:] += dx[m][:] in 2^-32 fixed point (int64 atomics: associative => deterministic)
__global__ void embed_scatter_kernel(const float* __restrict__ dx, const int* __restrict__ idx, long long* __restrict__ acc,
                                     long long M, int D) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * D) return;
  const long long m = i / D;
  const int c = (int)(i - m * D);
  const long long q = __double2ll_rn((double)dx[i] * 4294967296.0);
  atomicAdd(reinterpret_cast<unsigned long long*>(acc + (long long)idx[m] * D + c), (unsigned long long)q);
}
__global__ void fixed_to_float_kernel(const long long* __restrict__ acc, float* __restrict__ out, long long n, int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = (float)((double)acc[i] * (1.0 / 4294967296.0));
  out[i] = accumulate ? out[i] + v : v;
}

// nn.Dropout on a flat tensor, forward and backward alike: y[i] = x[i] * mask_i / (1 - p), mask from the counter hash
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n4, float p, float inv_keep, unsigned seed,
                               const unsigned* __restrict__ seed_dev) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  if (seed_dev) seed = *seed_dev;
  f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] *= sfmi_dropout_mul(seed, (unsigned)(4 * i + e), p, inv_keep);
  reinterpret_cast<f32x4*>(y)[i] = v;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

// AdamW (torch.optim.AdamW semantics, shapeformer.py:158-207): decoupled weight decay, bias-corrected moments
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}

// one launch for a whole parameter table: chunk c updates elements [coff[c], coff[c] + clen[c]) of tensor ctensor[c];
// gradient / moment buffers are flat (tensor t starts at foff[t]); weight decay is per tensor (AdamW parameter groups)
struct AdamMultiArgs {
  float* const* p; const long long* foff; const float* wd;
  const int* ctensor; const long long* coff; const int* clen;
  const float* g; float* m; float* v;
  float lr, b1, b2, eps, bc1, bc2;
  const float* bc_dev;   // optional {1 - beta1^t, 1 - beta2^t, lr} in device memory (the captured training step: t - and a scheduled lr - change between replays)
  float* pflat;   // optional: the updated parameter is ALSO written to pflat[flat index] (may alias g: the all-gather send buffer of the sharded update)
};
__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamMultiArgs a) {
  const int c = blockIdx.x, t = a.ctensor[c];
  const long long o = a.coff[c], fo = a.foff[t] + o;
  const int n = a.clen[c];
  float* p = a.p[t] + o;
  const float bc1 = a.bc_dev ? a.bc_dev[0] : a.bc1, bc2 = a.bc_dev ? a.bc_dev[1] : a.bc2, lr = a.bc_dev ? a.bc_dev[2] : a.lr;
  const float decay = 1.0f - lr * a.wd[t], sb2 = sqrtf(bc2), step = lr / bc1;
  // 9.1 GB of pure streaming per step at d = 1024 (read p, g, m, v; write p, m, v): 16-byte nontemporal accesses (the bare-stream
  // probe of round 5 reads 7.1 TB/s nontemporal against 6.4 TB/s with the default policy); the per-element arithmetic is unchanged
  if (((fo | o | (long long)n) & 3) == 0) {
    const f32x4* g4 = reinterpret_cast<const f32x4*>(a.g + fo);
    f32x4* m4 = reinterpret_cast<f32x4*>(a.m + fo);
    f32x4* v4 = reinterpret_cast<f32x4*>(a.v + fo);
    f32x4* p4 = reinterpret_cast<f32x4*>(p);
    f32x4* pf4 = a.pflat ? reinterpret_cast<f32x4*>(a.pflat + fo) : nullptr;
    for (int i = threadIdx.x; i < n / 4; i += 256) {
      const f32x4 gi = __builtin_nontemporal_load(g4 + i), mo = __builtin_nontemporal_load(m4 + i), vo = __builtin_nontemporal_load(v4 + i);
      const f32x4 po = __builtin_nontemporal_load(p4 + i);
      f32x4 mi, vi, pn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mi[e] = a.b1 * mo[e] + (1.0f - a.b1) * gi[e];
        vi[e] = a.b2 * vo[e] + (1.0f - a.b2) * gi[e] * gi[e];
        pn[e] = po[e] * decay - step * (mi[e] / (sqrtf(vi[e]) / sb2 + a.eps));
      }
      __builtin_nontemporal_store(mi, m4 + i);
      __builtin_nontemporal_store(vi, v4 + i);
      p4[i] = pn;
      if (pf4) pf4[i] = pn;
    }
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const float gi = a.g[fo + i];
    const float mi = a.b1 * a.m[fo + i] + (1.0f - a.b1) * gi;
    const float vi = a.b2 * a.v[fo + i] + (1.0f - a.b2) * gi * gi;
    a.m[fo + i] = mi; a.v[fo + i] = vi;
    const float pn = p[i] * decay - step * (mi / (sqrtf(vi) / sb2 + a.eps));
    p[i] = pn;
    if (a.pflat) a.pflat[fo + i] = pn;
  }
}

// flat buffer -> parameter tensors over the same chunk tables (after the all-gather of the sharded update)
__global__ __launch_bounds__(256) void unflatten_multi_kernel(float* const* p, const long long* foff, const int* ctensor, const long long* coff,
                                                              const int* clen, const float* __restrict__ flat) {
  const int c = blockIdx.x, t = ctensor[c];
  const long long o = coff[c], fo = foff[t] + o;
  const int n = clen[c];
  float* dst = p[t] + o;
  for (int i = threadIdx.x; i < n; i += 256) dst[i] = flat[fo + i];
}

extern "C" {

int sfmi_transpose_f32(const float* in, float* out, int R, int C, int ldin, int Rpad, void* stream) {
  if (!in || !out || R <= 0 || C <= 0 || Rpad < R) return SFMI_EINVAL;
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (Rpad + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, R, C, ldin, Rpad);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_colsum_f32(const float* x, float* out, int M, int N, int ld, int accumulate, void* stream) {
  if (!x || !out || M <= 0 || N <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, out, M, N, ld, accumulate);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// row slices used by the two-stage column reductions: enough (column block, slice) pairs to fill the chip
int sfmi_colsum_slices(int M, int N) {
  const int cb = (N + 63) / 64;
  int rs = (2048 + cb - 1) / cb;
  if (rs > 128) rs = 128;          // the finishing pass walks the slices serially per column
  if (rs > M / 16) rs = M / 16;
  return rs < 1 ? 1 : rs;
}
// same result contract as sfmi_colsum_f32 (fixed summation order), with scratch of sfmi_colsum_slices(M,N)*N floats
int sfmi_colsum_ws_f32(const float* x, float* out, int M, int N, int ld, int accumulate, float* scratch, void* stream) {
  if (!x || !out || !scratch || M <= 0 || N <= 0) return SFMI_EINVAL;
  const int RS = sfmi_colsum_slices(M, N), rows_per = (M + RS - 1) / RS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_part_kernel, dim3((N + 63) / 64, RS), dim3(256), 0, st, x, scratch, M, N, ld, rows_per);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((N + 255) / 256), dim3(256), 0, st, scratch, out, N, RS, accumulate);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_gelu_f32(const float* x, float* y, long long n, void* stream) {  // nn.GELU (mingpt.py:103)
  if (!x || !y || n <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_gelu_bwd_f32(const float* dy, const float* x, float* dx, long long n, void* stream) {
  if (!dy || !x || !dx || n <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// LayerNorm backward: dx = dLN/dx (+ dres) ; dgamma/dbeta ACCUMULATE.
// stats: scratch of sfmi_layernorm_bwd_scratch_floats(M, D) floats (row statistics + two-stage parameter partials).
size_t sfmi_layernorm_bwd_scratch_floats(int M, int D) { return (size_t)2 * M + (size_t)2 * D * sfmi_colsum_slices(M, D); }
int sfmi_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* dgamma,
                           float* dbeta, float* stats, int M, int D, void* stream) {
  if (!dy || !x || !gamma || !dx || !stats || M <= 0 || D <= 0) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3(M), dim3(256), 0, st, dy, x, gamma, dres, dx, stats, D);
  if (dgamma && dbeta) {
    const int RS = sfmi_colsum_slices(M, D), rows_per = (M + RS - 1) / RS;
    float* part = stats + (size_t)2 * M;
    hipLaunchKernelGGL(ln_bwd_params_part_kernel, dim3((D + 63) / 64, RS), dim3(256), 0, st, dy, x, stats, part, M, D, rows_per);
    hipLaunchKernelGGL(ln_bwd_params_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, st, part, dgamma, dbeta, D, RS);
  }
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// LayerNorm backward, row part only: dx = dLN/dx (+ dres), stats (M,2) = row mean / rstd for the parameter sums, which the caller
// adds to a block's sfmi_col_reduce_f32 launch (kind 1 job).
// drop_seed_dev != NULL: the seed is read from device memory at run time (the captured training step, csrc/train.hip header of this family)
int sfmi_layernorm_bwd_rows_drop_sd_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* stats,
                                        float* dx2, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, int M, int D, void* stream) {
  if (!dy || !x || !gamma || !dx || !stats || M <= 0 || D <= 0 || drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((M + 3) / 4);
  if (D == 1024) hipLaunchKernelGGL(ln_bwd_rows_wave_kernel<16>, grid, dim3(256), 0, st, dy, x, gamma, dres, dx, stats, dx2, M, drop_p, drop_seed, drop_seed_dev);
  else if (D == 512) hipLaunchKernelGGL(ln_bwd_rows_wave_kernel<8>, grid, dim3(256), 0, st, dy, x, gamma, dres, dx, stats, dx2, M, drop_p, drop_seed, drop_seed_dev);
  else if (D == 256) hipLaunchKernelGGL(ln_bwd_rows_wave_kernel<4>, grid, dim3(256), 0, st, dy, x, gamma, dres, dx, stats, dx2, M, drop_p, drop_seed, drop_seed_dev);
  else if (D == 128) hipLaunchKernelGGL(ln_bwd_rows_wave_kernel<2>, grid, dim3(256), 0, st, dy, x, gamma, dres, dx, stats, dx2, M, drop_p, drop_seed, drop_seed_dev);
  else {      // other widths: the block-per-row form, then the dropout as its own launch
    hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3(M), dim3(256), 0, st, dy, x, gamma, dres, dx, stats, D);
    if (dx2) {
      const long long n = (long long)M * D;
      if (n % 4) return SFMI_EINVAL;
      hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, dx, dx2, n / 4, drop_p, 1.0f / (1.0f - drop_p), drop_seed, drop_seed_dev);
    }
  }
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_layernorm_bwd_rows_drop_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* stats,
                                     float* dx2, float drop_p, unsigned drop_seed, int M, int D, void* stream) {
  return sfmi_layernorm_bwd_rows_drop_sd_f32(dy, x, gamma, dres, dx, stats, dx2, drop_p, drop_seed, nullptr, M, D, stream);
}
int sfmi_layernorm_bwd_rows_f32(const float* dy, const float* x, const float* gamma, const float* dres, float* dx, float* stats, int M,
                                int D, void* stream) {
  return sfmi_layernorm_bwd_rows_drop_f32(dy, x, gamma, dres, dx, stats, nullptr, 0.f, 0, M, D, stream);
}
// Up to 8 column reductions over M rows in one launch (csrc/train.hip:col_reduce_kernel).  Job i: kind[i] 0: out[i][n] (=|+=) sum_m
// a[i][m][n]  (bias gradient of a Linear layer);  kind[i] 1: out[i] = dgamma = sum_m a * (x - mean_m) * rstd_m, out2[i] = dbeta =
// sum_m a, stats[i] (M,2) from sfmi_layernorm_bwd_rows_f32.  a / x row stride ld[i], N[i] % 4 == 0.  part / cnt: scratch for tall
// inputs (sfmi_col_reduce_scratch: floats / ints; cnt zeroed ONCE).  Fixed summation order: deterministic.
int sfmi_col_reduce_slices(int M) { return M <= 1024 ? 1 : (M + 511) / 512 > 16 ? 16 : (M + 511) / 512; }
long long sfmi_col_reduce_part_floats(int M, int total_cols) { return (long long)((total_cols + 63) / 64 + CR_MAX_JOBS) * sfmi_col_reduce_slices(M) * 128; }
int sfmi_col_reduce_f32(int njobs, const int* kind, const float* const* a, const float* const* x, const float* const* stats, float* const* out,
                        float* const* out2, const int* N, const int* ld, int M, int accumulate, float* part, long long part_floats, int* cnt,
                        long long cnt_ints, void* stream) {
  if (njobs <= 0 || njobs > CR_MAX_JOBS || !kind || !a || !out || !N || !ld || M <= 0) return SFMI_EINVAL;
  ColJobs J;
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    if (!a[i] || !out[i] || N[i] <= 0 || N[i] % 4 || ld[i] % 4 || (kind[i] != 0 && kind[i] != 1)) return SFMI_EINVAL;
    if (kind[i] == 1 && (!x || !stats || !out2 || !x[i] || !stats[i] || !out2[i])) return SFMI_EINVAL;
    J.j[i].a = a[i]; J.j[i].x = x ? x[i] : nullptr; J.j[i].stats = stats ? stats[i] : nullptr; J.j[i].out = out[i];
    J.j[i].out2 = out2 ? out2[i] : nullptr; J.j[i].N = N[i]; J.j[i].ld = ld[i]; J.j[i].kind = kind[i]; J.j[i].blk0 = blk;
    blk += (N[i] + 63) / 64;
  }
  for (int i = njobs; i < CR_MAX_JOBS; ++i) J.j[i] = J.j[0];
  J.njobs = njobs; J.M = M; J.accumulate = accumulate; J.RS = sfmi_col_reduce_slices(M); J.part = part; J.cnt = cnt;
  if (J.RS > 1 && (!part || !cnt || (long long)blk * J.RS * 128 > part_floats || blk > cnt_ints)) return SFMI_EINVAL;
  hipLaunchKernelGGL(col_reduce_kernel, dim3(blk, J.RS), dim3(256), 0, (hipStream_t)stream, J);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// F.cross_entropy forward + backward over rows m=(b,t), active for t >= t0 (shapeformer.py:132-140)
int sfmi_ce_fwd_bwd_f32(const float* logits, const int* target, float* loss_rows, float* dlogits, int M, int V, int ld, int L,
                        int t0, float scale, void* stream) {
  if (!logits || !target || !loss_rows || !dlogits || M <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, target, loss_rows, dlogits, V, ld, L, t0, scale);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// causal self-attention backward (mingpt.py:73-91), head dim 64: dqkv (B*L,3D) from qkv, y, dy.  lse: 2*B*H*L floats of scratch
// (row log-sum-exps, then row sums of dO*O), both recomputed here (attn_stats_mfma_kernel) - the form for callers that did not keep
// the forward's log-sum-exps; then the fused dQ | dK/dV launch.
int sfmi_attn_bwd_f32(const float* qkv, const float* y, const float* dy, float* lse, float* dqkv, int B, int L, int D, int H,
                      float drop_p, unsigned drop_seed, void* stream) {
  if (!qkv || !y || !dy || !lse || !dqkv || D / H != 64 || D % H || drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nqb = (L + 63) / 64;
  float* delta = lse + (size_t)B * H * L;
  hipLaunchKernelGGL(attn_stats_mfma_kernel, dim3(B, H, nqb), dim3(256), 0, st, qkv, y, dy, lse, delta, L, D, 0.125f);
  attn_bwd_fused_launch(st, qkv, dy, lse, delta, dqkv, B, L, D, H, drop_p, drop_seed, nullptr);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// the same from the forward's (B,H,L) log-sum-exps (sfmi_gpt_attn_prefill_lse_f32): a row-sum launch for delta (B*H*L floats of scratch)
// + the fused launch.  H must divide 64 (head dim 64: D = 64 H).
int sfmi_attn_bwd_lse_sd_f32(const float* qkv, const float* y, const float* dy, const float* lse, float* delta, float* dqkv, int B, int L,
                             int D, int H, float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, void* stream) {
  if (!qkv || !y || !dy || !lse || !delta || !dqkv || H <= 0 || D != 64 * H || 64 % H || drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)(((long long)B * L + 3) / 4)), dim3(256), 0, st, y, dy, delta, B, L, D, H);
  attn_bwd_fused_launch(st, qkv, dy, lse, delta, dqkv, B, L, D, H, drop_p, drop_seed, drop_seed_dev);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_attn_bwd_lse_f32(const float* qkv, const float* y, const float* dy, const float* lse, float* delta, float* dqkv, int B, int L,
                          int D, int H, float drop_p, unsigned drop_seed, void* stream) {
  return sfmi_attn_bwd_lse_sd_f32(qkv, y, dy, lse, delta, dqkv, B, L, D, H, drop_p, drop_seed, nullptr, stream);
}
// Training forward of CausalSelfAttention (mingpt.py:73-91) for launches too small to fill the chip with the 64-row prefill tiles
// (B * H * ceil(L / 64) <= 128, head dim 64): 32-row tiles x two key-block groups per workgroup, (B,H,L) log-sum-exps for the backward
// pass.  Larger launches use sfmi_gpt_attn_prefill_lse_f32.  Returns SFMI_EINVAL when the launch is not "small".
int sfmi_attn_train_fwd_small_sd_f32(const float* qkv, float* y, float* lse, int B, int L, int D, int H, float drop_p, unsigned drop_seed,
                                     const unsigned* drop_seed_dev, void* stream) {
  if (!qkv || !y || !lse || B <= 0 || L <= 0 || H <= 0 || D != 64 * H || drop_p < 0.f || drop_p >= 1.f) return SFMI_EINVAL;
  if ((long long)B * H * ((L + 63) / 64) > 128) return SFMI_EINVAL;
  hipLaunchKernelGGL((attn_train_fwd_kernel<2, 2>), dim3(B, H, (L + 31) / 32), dim3(256), 0, (hipStream_t)stream, qkv, y, lse, L, D, 0.125f, drop_p,
                     drop_seed, drop_seed_dev);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_attn_train_fwd_small_f32(const float* qkv, float* y, float* lse, int B, int L, int D, int H, float drop_p, unsigned drop_seed,
                                  void* stream) {
  return sfmi_attn_train_fwd_small_sd_f32(qkv, y, lse, B, L, D, H, drop_p, drop_seed, nullptr, stream);
}
// nn.Embedding backward: acc (rows*D int64, zeroed by the caller) += scatter(dx) ; then sfmi_fixed_to_float_f32
int sfmi_embed_scatter_f32(const float* dx, const int* idx, long long* acc, long long M, int D, void* stream) {
  if (!dx || !idx || !acc || M <= 0) return SFMI_EINVAL;
  const long long n = M * D;
  hipLaunchKernelGGL(embed_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, idx, acc, M, D);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_fixed_to_float_f32(const long long* acc, float* out, long long n, int accumulate, void* stream) {
  if (!acc || !out || n <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc, out, n, accumulate);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// nn.Dropout(p) with the counter-hash mask of `seed` (embedding / residual dropouts and their backward: mingpt.py:90,105,218,292);
// n a multiple of 4; y may alias x
int sfmi_dropout_sd_f32(const float* x, float* y, long long n, float p, unsigned seed, const unsigned* seed_dev, void* stream) {
  if (!x || !y || n <= 0 || n % 4 || p < 0.f || p >= 1.f) return SFMI_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, p, 1.0f / (1.0f - p), seed,
                     seed_dev);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_dropout_f32(const float* x, float* y, long long n, float p, unsigned seed, void* stream) {
  return sfmi_dropout_sd_f32(x, y, n, p, seed, nullptr, stream);
}
int sfmi_add_f32(const float* a, const float* b, float* out, long long n, void* stream) {
  if (!a || !b || !out || n <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
// torch.optim.AdamW step on a flat parameter segment (shapeformer.py:158-207 groups: wd on Linear weights only)
int sfmi_adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return SFMI_EINVAL;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// the same update for a table of tensors in ONE launch.  Device tables: p (T pointers), foff (T flat offsets), wd (T),
// chunk tables ctensor / coff / clen (nchunks; any partition of every tensor into chunks, e.g. 16384 elements each).
// pflat (optional, may alias g): the updated parameters are also written to pflat[flat index] - the optimizer-sharded step
// (reduce-scatter -> this update on the rank's shard -> all-gather of pflat -> sfmi_unflatten_multi_f32) sends them from there.
// [host] the two bias corrections {1 - beta1^step, 1 - beta2^step} exactly as sfmi_adamw_multi_shard_f32 forms them
int sfmi_adamw_bias_corrections(float beta1, float beta2, int step, float* out2) {
  if (!out2 || step <= 0) return SFMI_EINVAL;
  out2[0] = 1.0f - powf(beta1, (float)step); out2[1] = 1.0f - powf(beta2, (float)step);
  return SFMI_OK;
}
// bc_dev != NULL: {bias correction 1, bias correction 2, learning rate} are read from device memory at run time (3 floats; the corrections as
// sfmi_adamw_bias_corrections forms them from the step count) instead of `step` / `lr` at launch time - the captured training step replays
// one launch for every step count and every scheduled learning rate
int sfmi_adamw_multi_shard_bc_f32(float* const* p, const long long* foff, const float* wd, const int* ctensor, const long long* coff,
                                  const int* clen, int nchunks, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                                  float eps, int step, const float* bc_dev, float* pflat, void* stream) {
  if (!p || !foff || !wd || !ctensor || !coff || !clen || !g || !m || !v || nchunks <= 0 || (step <= 0 && !bc_dev)) return SFMI_EINVAL;
  AdamMultiArgs a;
  a.pflat = pflat; a.bc_dev = bc_dev;
  a.p = p; a.foff = foff; a.wd = wd; a.ctensor = ctensor; a.coff = coff; a.clen = clen; a.g = g; a.m = m; a.v = v;
  a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
  a.bc1 = 1.0f - powf(beta1, (float)step); a.bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}
int sfmi_adamw_multi_shard_f32(float* const* p, const long long* foff, const float* wd, const int* ctensor, const long long* coff,
                               const int* clen, int nchunks, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                               float eps, int step, float* pflat, void* stream) {
  return sfmi_adamw_multi_shard_bc_f32(p, foff, wd, ctensor, coff, clen, nchunks, g, m, v, lr, beta1, beta2, eps, step, nullptr, pflat, stream);
}
int sfmi_adamw_multi_f32(float* const* p, const long long* foff, const float* wd, const int* ctensor, const long long* coff,
                         const int* clen, int nchunks, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                         float eps, int step, void* stream) {
  return sfmi_adamw_multi_shard_f32(p, foff, wd, ctensor, coff, clen, nchunks, g, m, v, lr, beta1, beta2, eps, step, nullptr, stream);
}
// parameter tensors <- flat buffer, chunk tables as in sfmi_adamw_multi_f32 (all-gathered parameters of the sharded update)
int sfmi_unflatten_multi_f32(float* const* p, const long long* foff, const int* ctensor, const long long* coff, const int* clen,
                             int nchunks, const float* flat, void* stream) {
  if (!p || !foff || !ctensor || !coff || !clen || !flat || nchunks <= 0) return SFMI_EINVAL;
  hipLaunchKernelGGL(unflatten_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, p, foff, ctensor, coff, clen, flat);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
