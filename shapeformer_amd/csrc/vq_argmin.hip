// Codebook nearest-neighbour (VQ) for gfx950: fused f32-MFMA GEMM + running argmin.
//
// Replaces Quantizer.forward's distance/argmax (reference shapeformer/models/vqdif/quantizer.py:47-51),
// which materialises an (N,4096) f32 distance matrix AND a same-size one-hot (64 MB + 64 MB per res16
// shape).  Here each wave keeps 64 latent rows in registers (MFMA B fragments), streams the fragment-
// packed codebook from L2, and folds  d = (|x|^2 - 2 x.w) + |w|^2  (same expression order as the
// reference) into a per-lane running (min, index); nothing but the int32 index is written.
// Tie rule: lowest code index (CPU torch.max first-max semantics, quantizer.py:51).
#include "sfmi_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// codebook pack: [K/32 tiles][D/8 subs][64 lanes][4]  + |w|^2 [K]
template <int D>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ x,      // (N,D) row-major
                                                        const float* __restrict__ wpack,  // packed codebook
                                                        const float* __restrict__ ww,     // (K)
                                                        int* __restrict__ idx_out,        // (N)
                                                        float* __restrict__ dmin_out,     // (N) optional
                                                        long long N, int K) {
  constexpr int NS = D / 8;
  const int lane = threadIdx.x & 63, hi = lane >> 5, pl = lane & 31;
  const long long wave_gid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long row0 = wave_gid * 64;
  if (row0 >= N) return;

  f32x4 xf[2][NS];
  float xx[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    long long r = row0 + v * 32 + pl;
    if (r >= N) r = N - 1;
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + r * D + 4 * hi);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      xf[v][k] = xp[2 * k];
      s += xf[v][k][0] * xf[v][k][0] + xf[v][k][1] * xf[v][k][1] + xf[v][k][2] * xf[v][k][2] +
           xf[v][k][3] * xf[v][k][3];
    }
    xx[v] = s + __shfl_xor(s, 32, 64);
  }
  float best[2] = {INFINITY, INFINITY};
  int bidx[2] = {0, 0};
  const f32x4* wp = reinterpret_cast<const f32x4*>(wpack) + lane;
  const int ntiles = K / 32;
  for (int ct = 0; ct < ntiles; ++ct) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc0[t] = 0.0f; acc1[t] = 0.0f; }
    const f32x4* wt = wp + (long long)ct * NS * 64;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      f32x4 w = wt[k * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc0 = MFMA(w[j], xf[0][k][j], acc0);
        acc1 = MFMA(w[j], xf[1][k][j], acc1);
      }
    }
    const f32x4* wwp = reinterpret_cast<const f32x4*>(ww + ct * 32 + 4 * hi);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w2 = wwp[2 * g];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int code = ct * 32 + 8 * g + 4 * hi + j;
        float d0 = (xx[0] - 2.0f * acc0[4 * g + j]) + w2[j];
        float d1 = (xx[1] - 2.0f * acc1[4 * g + j]) + w2[j];
        if (d0 < best[0]) { best[0] = d0; bidx[0] = code; }
        if (d1 < best[1]) { best[1] = d1; bidx[1] = code; }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    float ob = __shfl_xor(best[v], 32, 64);
    int oi = __shfl_xor(bidx[v], 32, 64);
    if (ob < best[v] || (ob == best[v] && oi < bidx[v])) { best[v] = ob; bidx[v] = oi; }
    long long r = row0 + v * 32 + pl;
    if (hi == 0 && r < N) {
      idx_out[r] = bidx[v];
      if (dmin_out) dmin_out[r] = best[v];
    }
  }
}

// get_code (quantizer.py:19-30): W[idx] rows; channels-last output (N,D) == (B,R,R,R,D)
__global__ void vq_gather_kernel(const float* __restrict__ W, const int* __restrict__ idx, float* __restrict__ out,
                                 long long N, int D) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = D / 4;
  long long r = gid / per;
  int c = (int)(gid - r * per);
  if (r >= N) return;
  reinterpret_cast<f32x4*>(out + r * D)[c] = reinterpret_cast<const f32x4*>(W + (long long)idx[r] * D)[c];
}

extern "C" {

size_t sfmi_vq_pack_floats(int K, int D) { return (size_t)K * D + K; }

// host packer: embedding.weight (K,D) -> fragment-ordered codebook followed by |w|^2 (K)
int sfmi_vq_pack_codebook(const float* W, int K, int D, float* out) {
  if (!W || !out || K % 32 || D % 8) return SFMI_EINVAL;
  const int NS = D / 8;
  for (int ct = 0; ct < K / 32; ++ct)
    for (int s = 0; s < NS; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j)
          out[(((size_t)ct * NS + s) * 64 + l) * 4 + j] = W[(size_t)(ct * 32 + (l & 31)) * D + 8 * s + 4 * (l >> 5) + j];
  float* ww = out + (size_t)K * D;
  for (int k = 0; k < K; ++k) {
    float s = 0.0f;
    for (int c = 0; c < D; ++c) s += W[(size_t)k * D + c] * W[(size_t)k * D + c];
    ww[k] = s;
  }
  return SFMI_OK;
}

// replaces Quantizer.forward distances+argmax (quantizer.py:47-51). x: (N,D) channels-last latent rows.
int sfmi_vq_argmin_f32(const float* x, const float* packed, int* idx_out, float* dmin_out, long long N, int K, int D,
                       void* stream) {
  if (!x || !packed || !idx_out || N <= 0 || K % 32 || (D != 64 && D != 128)) return SFMI_EINVAL;
  const float* ww = packed + (size_t)K * D;
  long long waves = (N + 63) / 64;
  dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  if (D == 128)
    hipLaunchKernelGGL(vq_argmin_kernel<128>, grid, block, 0, (hipStream_t)stream, x, packed, ww, idx_out, dmin_out, N, K);
  else
    hipLaunchKernelGGL(vq_argmin_kernel<64>, grid, block, 0, (hipStream_t)stream, x, packed, ww, idx_out, dmin_out, N, K);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces Quantizer.get_code (quantizer.py:19-30); output is channels-last (N,D).
int sfmi_vq_gather_f32(const float* W, const int* idx, float* out, long long N, int D, void* stream) {
  if (!W || !idx || !out || N <= 0 || D % 4) return SFMI_EINVAL;
  long long n = N * (D / 4);
  hipLaunchKernelGGL(vq_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, idx, out, N, D);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
