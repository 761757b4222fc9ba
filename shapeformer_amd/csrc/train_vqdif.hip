// Training kernels of the VQDIF autoencoder: the backward halves of the encoder's local pooling,
// the 3-D convolutions / GroupNorm / pooling of the down- and up-samplers and the UNet, the trilinear feature sampling
// of the implicit decoder, the BCE loss and the EMA codebook update (vqdif.py:78-137, quantizer.py:68-86, enc.py:66-140,
// unet3d.py, updown.py, dec.py:62-100).  Forward passes in training reuse the inference kernels where they apply
// (conv3d_igemm, sfmi_gemm_f32, GroupNorm coefficients, max-pool); GEMM-shaped input gradients reuse the forward
// kernels on transposed / tap-flipped weights.  Everything that sums over points or voxels in a data-dependent order
// goes through 2^-32 fixed-point int64 atomics, so gradients are bit-reproducible run to run.
#include "sfmi_common.h"
#include <limits.h>

namespace {

constexpr double FIX_SCALE = 4294967296.0;
__device__ __forceinline__ long long to_fix(float v) { return __double2ll_rn((double)v * FIX_SCALE); }
__device__ __forceinline__ void atomic_add_ll(long long* p, long long v) { atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); }
// monotone float <-> int key (signed int order == float order)
__device__ __forceinline__ int f2key(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// ------------------------------------------------------------------------------------------------ elementwise
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 d = reinterpret_cast<const f32x4*>(dy)[i], v = reinterpret_cast<const f32x4*>(y)[i];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? d[e] : 0.f;
  reinterpret_cast<f32x4*>(dx)[i] = o;
}

__global__ void lincomb_kernel(float a, const float* __restrict__ x, float b, const float* __restrict__ y, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}

// out = A[b,c] u + Bc[b,c] v + Cc[b,c]   (GroupNorm backward is affine in (dy, x) per (sample, channel))
__global__ void affine2_kernel(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ A,
                               const float* __restrict__ Bc, const float* __restrict__ Cc, float* __restrict__ out, long long V,
                               int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long b = i / ((long long)V * C);
  const long long k = b * C + c;
  out[i] = A[k] * u[i] + Bc[k] * v[i] + Cc[k];
}

// per (b, c): sum_v dy, sum_v dy*x  -> partial (B, S, C, 2) doubles.  256 threads = (256/C4) voxel lanes x C4 float4
// channel groups (C4 = C/4 <= 256), LDS tree over the voxel lanes.
__global__ __launch_bounds__(256) void chan_dot_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             double* __restrict__ partial, long long V, int C, int S) {
  __shared__ double red[256 * 8];
  const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cg = C / 4, rpp = 256 / cg;
  const long long v0 = V * s / S, v1 = V * (s + 1) / S;
  const int g = tid % cg, ro = tid / cg;
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  if (ro < rpp)
    for (long long v = v0 + ro; v < v1; v += rpp) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(dy + ((long long)b * V + v) * C + 4 * g);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((long long)b * V + v) * C + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += (double)d[e]; a1[e] += (double)d[e] * (double)xv[e]; }
    }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = a0[e]; red[tid * 8 + 4 + e] = a1[e]; }
  __syncthreads();
  if (ro == 0) {
    for (int r = 1; r < rpp; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += red[(tid + r * cg) * 8 + e]; a1[e] += red[(tid + r * cg) * 8 + 4 + e]; }
    double* o = partial + (((long long)b * S + s) * C + 4 * g) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = a0[e]; o[2 * e + 1] = a1[e]; }
  }
}

// ------------------------------------------------------------------------------------------------ pooling / resampling
__global__ void upsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int D, int H, int W, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (B, 2D, 2H, 2W, C)
  if (i >= total) return;
  const int c = (int)(i % C);
  long long r = i / C;
  const int w = (int)(r % (2 * W)); r /= 2 * W;
  const int h = (int)(r % (2 * H)); r /= 2 * H;
  const int d = (int)(r % (2 * D));
  const long long b = r / (2 * D);
  y[i] = x[((((b * D + (d >> 1)) * H + (h >> 1)) * W + (w >> 1))) * C + c];
}

// dx[b,o,c] = sum over the 2x2x2 children of dy[b, 2o+t, c0 + c]   (backward of nearest x2 upsampling on a channel slice)
__global__ void sumpool2_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Do, int Ho, int Wo, int Ct, int c0, int Cs,
                                long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (B, Do, Ho, Wo, Cs)
  if (i >= total) return;
  const int c = (int)(i % Cs);
  long long r = i / Cs;
  const int w = (int)(r % Wo); r /= Wo;
  const int h = (int)(r % Ho); r /= Ho;
  const int d = (int)(r % Do);
  const long long b = r / Do;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const long long v = ((b * 2 * Do + 2 * d + (t >> 2)) * 2 * Ho + 2 * h + ((t >> 1) & 1)) * 2 * Wo + 2 * w + (t & 1);
    s += dy[v * Ct + c0 + c];
  }
  dx[i] = s;
}

// max_pool3d(2) backward: the gradient goes to the FIRST child (z-major scan order, as ATen) equal to the pooled value
__global__ void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                    float* __restrict__ dx, int Do, int Ho, int Wo, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (B, Do, Ho, Wo, C)
  if (i >= total) return;
  const int c = (int)(i % C);
  long long r = i / C;
  const int w = (int)(r % Wo); r /= Wo;
  const int h = (int)(r % Ho); r /= Ho;
  const int d = (int)(r % Do);
  const long long b = r / Do;
  const float m = y[i], g = dy[i];
  bool done = false;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const long long v = ((b * 2 * Do + 2 * d + (t >> 2)) * 2 * Ho + 2 * h + ((t >> 1) & 1)) * 2 * Wo + 2 * w + (t & 1);
    const bool hit = !done && x[v * C + c] == m;
    dx[v * C + c] = hit ? g : 0.f;
    done |= hit;
  }
}

// ------------------------------------------------------------------------------------------------ encoder pooling
// cell index of every point (vqdif.py:36 cloud/2, vqdif/common.py:260-321), G^3 grid, 'original' order x + G (y + G z)
__global__ void cells_kernel(const float* __restrict__ cloud, int* __restrict__ cell, float* __restrict__ p_half, long long n, int G) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = cloud + i * 3;
  const float hx = p[0] * 0.5f, hy = p[1] * 0.5f, hz = p[2] * 0.5f;
  const int cx = (int)(sfmi_normalize(hx) * (float)G), cy = (int)(sfmi_normalize(hy) * (float)G), cz = (int)(sfmi_normalize(hz) * (float)G);
  cell[i] = cx + G * (cy + G * cz);
  if (p_half) { p_half[i * 3] = hx; p_half[i * 3 + 1] = hy; p_half[i * 3 + 2] = hz; }
}

__global__ void cell_max_scatter_kernel(const float* __restrict__ net, const int* __restrict__ cell, int* __restrict__ keys, int T,
                                        long long ncell, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (B, T, C)
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pt = i / C, b = pt / T;
  atomicMax(&keys[(b * ncell + cell[pt]) * C + c], f2key(net[i]));
}

// out[pt][co + c] = max of the point's cell  (written into the right half of the (B,T,ldo) concat buffer)
__global__ void cell_gather_max_kernel(const int* __restrict__ keys, const int* __restrict__ cell, float* __restrict__ out, int T,
                                       long long ncell, int C, int ldo, int co, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pt = i / C, b = pt / T;
  out[pt * ldo + co + c] = key2f(keys[(b * ncell + cell[pt]) * C + c]);
}

// acc[b, cell, c] += src[pt][cs + c]  (fixed point)
__global__ void cell_scatter_add_kernel(const float* __restrict__ src, const int* __restrict__ cell, long long* __restrict__ acc,
                                        int* __restrict__ count, int T, long long ncell, int C, int lds_, int cs, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pt = i / C, b = pt / T;
  const long long g = b * ncell + cell[pt];
  atomic_add_ll(&acc[g * C + c], to_fix(src[pt * lds_ + cs + c]));
  if (count && c == 0) atomicAdd(&count[g], 1);
}

// local max-pool backward: dnet[pt][c] (+)= (net[pt][c] == cell max) ? d_cell[cell][c] : 0
__global__ void cell_max_bwd_kernel(const float* __restrict__ net, const int* __restrict__ keys, const long long* __restrict__ acc,
                                    const int* __restrict__ cell, float* __restrict__ dnet, int T, long long ncell, int C, int ldd,
                                    int accumulate, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pt = i / C, b = pt / T;
  const long long g = (b * ncell + cell[pt]) * C + c;
  const float v = (f2key(net[i]) == keys[g]) ? (float)((double)acc[g] * (1.0 / FIX_SCALE)) : 0.f;
  float* o = dnet + pt * ldd + c;
  *o = accumulate ? *o + v : v;
}

// scatter_mean forward: grid[b,cell,c] = acc / max(count,1) for every cell (dense, zeros where empty)
__global__ void cell_mean_kernel(const long long* __restrict__ acc, const int* __restrict__ count, float* __restrict__ grid, int C,
                                 long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (B, ncell, C)
  if (i >= total) return;
  const int n = count[i / C];
  grid[i] = n ? (float)((double)acc[i] * (1.0 / FIX_SCALE) / (double)n) : 0.f;
}

// scatter_mean backward: dc[pt][c] = dgrid[b, cell, c] / count
__global__ void cell_mean_bwd_kernel(const float* __restrict__ dgrid, const int* __restrict__ count, const int* __restrict__ cell,
                                     float* __restrict__ dc, int T, long long ncell, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pt = i / C, b = pt / T;
  const long long g = b * ncell + cell[pt];
  dc[i] = dgrid[g * C + c] / (float)count[g];
}

// ------------------------------------------------------------------------------------------------ trilinear sampling
struct Axis { int i0, i1; float w0, w1; };
// grid_sample(align_corners=True, border) of one coordinate; p_api in [-1,1] (dec.py:62-68 after vqdif.py:71 Xtg/2)
__device__ __forceinline__ Axis tri_axis(float x_api, int G) {
  const float u = sfmi_normalize(x_api * 0.5f);
  const float v = 2.0f * u - 1.0f;
  float ix = ((v + 1.0f) / 2.0f) * (float)(G - 1);
  ix = fminf((float)(G - 1), fmaxf(ix, 0.0f));
  const float f0 = floorf(ix);
  Axis a;
  a.i0 = (int)f0; a.i1 = min(a.i0 + 1, G - 1); a.w1 = ix - f0; a.w0 = (f0 + 1.0f) - ix;
  return a;
}

// BWD == false: out[pt][c] = sum_corners w grid[corner][c];  BWD == true: acc[corner][c] += w dout[pt][c] (fixed point)
template <bool BWD>
__global__ void trilinear_kernel(const float* __restrict__ xyz, const float* __restrict__ grid, float* __restrict__ out,
                                 const float* __restrict__ dout, long long* __restrict__ acc, long long N, int G, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (B, N, C)
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long pt = i / C, b = pt / N;
  const float* p = xyz + pt * 3;
  const Axis ax = tri_axis(p[0], G), ay = tri_axis(p[1], G), az = tri_axis(p[2], G);
  const long long base = b * G * G * G;
  float s = 0.f;
  const float d = BWD ? dout[i] : 0.f;
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int dz = corner >> 2, dy = (corner >> 1) & 1, dx = corner & 1;
    const int zi = dz ? az.i1 : az.i0, yi = dy ? ay.i1 : ay.i0, xi = dx ? ax.i1 : ax.i0;
    const float w = ((dx ? ax.w1 : ax.w0) * (dy ? ay.w1 : ay.w0)) * (dz ? az.w1 : az.w0);
    const long long g = (base + ((long long)zi * G + yi) * G + xi) * C + c;
    if (BWD) atomic_add_ll(&acc[g], to_fix(w * d));
    else s = fmaf(grid[g], w, s);
  }
  if (!BWD) out[i] = s;
}

// ------------------------------------------------------------------------------------------------ loss
// nn.BCEWithLogitsLoss (mean): row loss = max(x,0) - x y + log1p(exp(-|x|)); dlogit = (sigmoid(x) - y) * scale
__global__ void bce_logits_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ loss,
                                  float* __restrict__ dx, long long n, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i], t = y[i];
  loss[i] = fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
  const float sg = 1.0f / (1.0f + expf(-v));
  dx[i] = (sg - t) * scale;
}

// ------------------------------------------------------------------------------------------------ quantizer EMA
// counts[k] += 1, sums[k][:] += x[row][:] for k = idx[row]  (quantizer.py:70-74: onehot.sum(0), inputs^T onehot)
__global__ void vq_stats_kernel(const float* __restrict__ x, const int* __restrict__ idx, long long* __restrict__ sums,
                                int* __restrict__ counts, int D, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (rows, D)
  if (i >= total) return;
  const int d = (int)(i % D);
  const long long row = i / D;
  const int k = idx[row];
  atomic_add_ll(&sums[(long long)k * D + d], to_fix(x[i]));
  if (d == 0) atomicAdd(&counts[k], 1);
}

// N = g N + (1-g) counts ; z = g z + (1-g) sums ; n = sum N ; emb = z / ((N + eps) / (n + K eps) * n)   (quantizer.py:70-86)
__global__ __launch_bounds__(1024) void vq_ema_kernel(float* __restrict__ Nbuf, float* __restrict__ z, float* __restrict__ emb,
                                                      const float* __restrict__ counts /*all-reduced, float*/,
                                                      const float* __restrict__ sums /*all-reduced, float*/, int K, int D, float gamma,
                                                      float eps) {
  __shared__ float red[1024];
  float part = 0.f;
  for (int k = threadIdx.x; k < K; k += 1024) {
    const float v = Nbuf[k] * gamma + (1.0f - gamma) * counts[k];
    Nbuf[k] = v;
    part += v;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  const float n = red[0];
  for (long long i = threadIdx.x; i < (long long)K * D; i += 1024) {
    const int k = (int)(i / D);
    const float zz = z[i] * gamma + (1.0f - gamma) * sums[i];
    z[i] = zz;
    const float wgt = (Nbuf[k] + eps) / (n + (float)K * eps) * n;
    emb[i] = zz / wgt;
  }
}

// ------------------------------------------------------------------------------------------------ conv weight gradient
// dW[tap][co][ci] = sum over output rows r of dY[r][co] * Xshift_tap[r][ci]   (channels-last; zero padding; stride)
// Block = 64 co x 64 ci tile of one tap over one slice of the rows (split-K); 4 waves, each a 32x32 sub-tile of
// 16x16x4 f32 MFMAs; rows are staged 16 at a time through LDS.  part: (nsplit, taps, Cout, Cin); reduce with colsum.
struct WgArgs {
  const float* dy; const float* x; float* part;
  int B, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, KS, stride, pad, ldy, ldx;
  long long rows, rows_per_split;
};
constexpr int WG_LD = 80;   // LDS row stride (floats): the 4 k-rows of one MFMA operand land in disjoint banks

__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgArgs a) {
  __shared__ float As[16][WG_LD], Bs[16][WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci_tiles = (a.Cin + 63) / 64;
  const int co0 = (blockIdx.x / ci_tiles) * 64, ci0 = (blockIdx.x % ci_tiles) * 64;
  const int tap = blockIdx.y;
  const int tz = tap / (a.KS * a.KS), ty = (tap / a.KS) % a.KS, tx = tap % a.KS;
  const long long r0 = (long long)blockIdx.z * a.rows_per_split, r1 = min(a.rows, r0 + a.rows_per_split);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;   // this wave's sub-tile origin (co, ci) inside the block tile
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int col = tid & 63, rsub = tid >> 6;
  const bool co_ok = co0 + col < a.Cout, ci_ok = ci0 + col < a.Cin;
  for (long long rb = r0; rb < r1; rb += 16) {
    float va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long r = rb + rsub + 4 * i;
      va[i] = 0.f; vb[i] = 0.f;
      if (r < r1) {
        if (co_ok) va[i] = a.dy[r * a.ldy + co0 + col];
        long long q = r;
        const int ox = (int)(q % a.Wo); q /= a.Wo;
        const int oy = (int)(q % a.Ho); q /= a.Ho;
        const int oz = (int)(q % a.Do);
        const long long b = q / a.Do;
        const int iz = oz * a.stride + tz - a.pad, iy = oy * a.stride + ty - a.pad, ix = ox * a.stride + tx - a.pad;
        if (ci_ok && iz >= 0 && iz < a.Di && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi)
          vb[i] = a.x[(((b * a.Di + iz) * a.Hi + iy) * a.Wi + ix) * a.ldx + ci0 + col];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) { As[rsub + 4 * i][col] = va[i]; Bs[rsub + 4 * i][col] = vb[i]; }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = kk * 4 + (lane >> 4);
      const float a0 = As[k][wm + (lane & 15)], a1 = As[k][wm + 16 + (lane & 15)];
      const float b0 = Bs[k][wn + (lane & 15)], b1 = Bs[k][wn + 16 + (lane & 15)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // C/D layout of 16x16x4: register j of lane l is row 4 (l >> 4) + j, column l & 15
  const long long taps = (long long)a.KS * a.KS * a.KS;
  float* out = a.part + (((long long)blockIdx.z * taps + tap) * a.Cout) * a.Cin;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + wm + 16 * i + 4 * (lane >> 4) + e, ci = ci0 + wn + 16 * j + (lane & 15);
        if (co < a.Cout && ci < a.Cin) out[(long long)co * a.Cin + ci] = acc[i][j][e];
      }
}

inline unsigned nblk(long long n, int t = 256) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" {

int sfmi_relu_bwd_f32(const float* dy, const float* y, float* dx, long long n, void* stream) {
  if (!dy || !y || !dx || n % 4) return SFMI_EINVAL;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblk(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n / 4);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_lincomb_f32(float a, const float* x, float b, const float* y, float* out, long long n, void* stream) {
  if (!x || !out) return SFMI_EINVAL;
  hipLaunchKernelGGL(lincomb_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, a, x, b, y, out, n);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_affine2_cl_f32(const float* u, const float* v, const float* A, const float* Bc, const float* Cc, float* out, int B,
                        long long V, int C, void* stream) {
  if (!u || !v || !A || !Bc || !Cc || !out) return SFMI_EINVAL;
  const long long total = (long long)B * V * C;
  hipLaunchKernelGGL(affine2_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, u, v, A, Bc, Cc, out, V, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* partial: B*S*C*2 doubles with S = sfmi_gn_splits(V) */
int sfmi_chan_dot_stats_f32(const float* dy, const float* x, double* partial, int B, long long V, int C, int S, void* stream) {
  if (!dy || !x || !partial || S <= 0 || C % 4 || C > 1024) return SFMI_EINVAL;
  hipLaunchKernelGGL(chan_dot_stats_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, dy, x, partial, V, C, S);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_upsample2_cl_f32(const float* x, float* y, int B, int D, int H, int W, int C, void* stream) {
  if (!x || !y) return SFMI_EINVAL;
  const long long total = (long long)B * 8 * D * H * W * C;
  hipLaunchKernelGGL(upsample2_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, D, H, W, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_sumpool2_cl_f32(const float* dy, float* dx, int B, int Do, int Ho, int Wo, int Ctot, int c0, int Cs, void* stream) {
  if (!dy || !dx || c0 + Cs > Ctot) return SFMI_EINVAL;
  const long long total = (long long)B * Do * Ho * Wo * Cs;
  hipLaunchKernelGGL(sumpool2_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, Do, Ho, Wo, Ctot, c0, Cs, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_maxpool2_bwd_cl_f32(const float* x, const float* y, const float* dy, float* dx, int B, int Do, int Ho, int Wo, int C,
                             void* stream) {
  if (!x || !y || !dy || !dx) return SFMI_EINVAL;
  const long long total = (long long)B * Do * Ho * Wo * C;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, y, dy, dx, Do, Ho, Wo, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_cells_f32(const float* cloud, int* cell, float* p_half, int B, int T, int G, void* stream) {
  if (!cloud || !cell) return SFMI_EINVAL;
  const long long n = (long long)B * T;
  hipLaunchKernelGGL(cells_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, cloud, cell, p_half, n, G);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* keys (B,ncell,C) int32 must be pre-filled with INT_MIN (hipMemset 0x80); out row stride ldo, column offset co */
int sfmi_cell_max_f32(const float* net, const int* cell, int* keys, float* out, int B, int T, long long ncell, int C, int ldo, int co,
                      void* stream) {
  if (!net || !cell || !keys || !out) return SFMI_EINVAL;
  const long long total = (long long)B * T * C;
  hipLaunchKernelGGL(cell_max_scatter_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, net, cell, keys, T, ncell, C, total);
  hipLaunchKernelGGL(cell_gather_max_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, keys, cell, out, T, ncell, C, ldo, co, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* acc (B,ncell,C) int64 zeroed by the caller; src row stride lds, column offset cs; count (B,ncell) optional */
int sfmi_cell_scatter_add_f32(const float* src, const int* cell, long long* acc, int* count, int B, int T, long long ncell, int C,
                              int lds, int cs, void* stream) {
  if (!src || !cell || !acc) return SFMI_EINVAL;
  const long long total = (long long)B * T * C;
  hipLaunchKernelGGL(cell_scatter_add_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, src, cell, acc, count, T, ncell, C, lds, cs, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_cell_max_bwd_f32(const float* net, const int* keys, const long long* acc, const int* cell, float* dnet, int B, int T,
                          long long ncell, int C, int ldd, int accumulate, void* stream) {
  if (!net || !keys || !acc || !cell || !dnet) return SFMI_EINVAL;
  const long long total = (long long)B * T * C;
  hipLaunchKernelGGL(cell_max_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, net, keys, acc, cell, dnet, T, ncell, C, ldd, accumulate, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_cell_mean_f32(const long long* acc, const int* count, float* grid, int B, long long ncell, int C, void* stream) {
  if (!acc || !count || !grid) return SFMI_EINVAL;
  const long long total = (long long)B * ncell * C;
  hipLaunchKernelGGL(cell_mean_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, acc, count, grid, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_cell_mean_bwd_f32(const float* dgrid, const int* count, const int* cell, float* dc, int B, int T, long long ncell, int C,
                           void* stream) {
  if (!dgrid || !count || !cell || !dc) return SFMI_EINVAL;
  const long long total = (long long)B * T * C;
  hipLaunchKernelGGL(cell_mean_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, dgrid, count, cell, dc, T, ncell, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* xyz (B,N,3) in [-1,1]; grid (B,G,G,G,C) -> out (B,N,C) */
int sfmi_trilinear_cl_f32(const float* xyz, const float* grid, float* out, int B, long long N, int G, int C, void* stream) {
  if (!xyz || !grid || !out) return SFMI_EINVAL;
  const long long total = (long long)B * N * C;
  hipLaunchKernelGGL((trilinear_kernel<false>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, xyz, grid, out, nullptr, nullptr, N, G, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* acc (B,G,G,G,C) int64 zeroed by the caller; convert with sfmi_fixed_to_float_f32 */
int sfmi_trilinear_bwd_cl_f32(const float* xyz, const float* dout, long long* acc, int B, long long N, int G, int C, void* stream) {
  if (!xyz || !dout || !acc) return SFMI_EINVAL;
  const long long total = (long long)B * N * C;
  hipLaunchKernelGGL((trilinear_kernel<true>), dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, xyz, nullptr, nullptr, dout, acc, N, G, C, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_bce_logits_f32(const float* logits, const float* label, float* loss_rows, float* dlogits, long long n, float scale,
                        void* stream) {
  if (!logits || !label || !loss_rows || !dlogits) return SFMI_EINVAL;
  hipLaunchKernelGGL(bce_logits_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, logits, label, loss_rows, dlogits, n, scale);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* sums (K,D) int64 and counts (K) int32 zeroed by the caller */
int sfmi_vq_stats_f32(const float* x, const int* idx, long long* sums, int* counts, long long rows, int D, void* stream) {
  if (!x || !idx || !sums || !counts) return SFMI_EINVAL;
  const long long total = rows * D;
  hipLaunchKernelGGL(vq_stats_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, idx, sums, counts, D, total);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_vq_ema_update_f32(float* N, float* z_avg, float* emb, const float* counts, const float* sums, int K, int D, float gamma,
                           float eps, void* stream) {
  if (!N || !z_avg || !emb || !counts || !sums) return SFMI_EINVAL;
  hipLaunchKernelGGL(vq_ema_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, N, z_avg, emb, counts, sums, K, D, gamma, eps);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

/* Weight gradient of a channels-last Conv3d / Linear (KS = 1): part (nsplit, KS^3, Cout, Cin) partial sums over
 * row slices of `rows_per_split` output voxels (multiple of 16); reduce over nsplit with sfmi_colsum_f32. */
int sfmi_conv3d_wgrad_f32(const float* dy, const float* x, float* part, int B, int Di, int Hi, int Wi, int Cin, int Cout, int KS,
                          int stride, int pad, int ldy, int ldx, int nsplit, void* stream) {
  if (!dy || !x || !part || nsplit <= 0 || KS <= 0 || stride <= 0) return SFMI_EINVAL;
  WgArgs a;
  a.dy = dy; a.x = x; a.part = part; a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KS = KS; a.stride = stride;
  a.pad = pad; a.ldy = ldy; a.ldx = ldx;
  a.Do = (Di + 2 * pad - KS) / stride + 1; a.Ho = (Hi + 2 * pad - KS) / stride + 1; a.Wo = (Wi + 2 * pad - KS) / stride + 1;
  a.rows = (long long)B * a.Do * a.Ho * a.Wo;
  a.rows_per_split = ((a.rows + nsplit - 1) / nsplit + 15) / 16 * 16;
  dim3 grid(((Cout + 63) / 64) * ((Cin + 63) / 64), KS * KS * KS, nsplit);
  hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
