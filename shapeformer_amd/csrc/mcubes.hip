// Iso-surface extraction on the GPU: marching cubes over the (B,Q,Q,Q) occupancy grid that the SDF
// query leaves in HBM, so "128^3 SDF extract" ends in an indexed triangle mesh without the 8 MB D2H + CPU PyMCubes step
// of the reference (xgutils/geoutil.py:175-233 array2mesh -> mcubes.marching_cubes, thresh .5, called from
// shapeformer.py:355-356 vis_ind and npfvis.plot_3d_recon).
//
// Indexed mesh by construction: every vertex lies on one grid edge (grid point p, axis a), so
//   pass 1  count cut edges / triangles per element and exclusive-scan both (three-kernel scan, counts recomputed
//           from the grid instead of materialised);
//   pass 2  write vertex vid[3p+a] at p + t e_a, t = (iso - f0)/(f1 - f0), mapped to the bounding box exactly like
//           array2mesh (verts/(Q-1)*(bbmax-bbmin)+bbmin); write triangles as vid lookups through the generated case
//           table (mc_table.h, built by shapeformer_amd/mc_tables.py).
// Output order is deterministic: vertices by (grid point, axis), triangles by (cell, table order).
// HBM-bound integer/byte work: ~9 reads of the grid (L2-resident 8 MB per 128^3 shape) + one int per edge and cell.
#include "sfmi_common.h"
#include "mc_table.h"

namespace {

struct McGrid {
  const float* occ; float iso; int B, Q;
  __device__ __forceinline__ long long npts() const { return (long long)Q * Q * Q; }
};

// 1 if grid edge (point p of shape b, axis a) is cut
struct EdgeCount {
  McGrid g;
  __device__ __forceinline__ int operator()(long long i) const {
    const long long n3 = g.npts();
    const int a = (int)(i % 3);
    const long long pp = i / 3;
    const long long p = pp % n3;
    const int i2 = (int)(p % g.Q), i1 = (int)((p / g.Q) % g.Q), i0 = (int)(p / ((long long)g.Q * g.Q));
    const int c = a == 0 ? i0 : (a == 1 ? i1 : i2);
    if (c + 1 >= g.Q) return 0;
    const long long step = a == 0 ? (long long)g.Q * g.Q : (a == 1 ? g.Q : 1);
    const float f0 = g.occ[pp], f1 = g.occ[pp + step];
    return (f0 > g.iso) != (f1 > g.iso);
  }
};

__device__ __forceinline__ int mc_cube_index(const McGrid& g, long long pp) {
  const long long s0 = (long long)g.Q * g.Q, s1 = g.Q;
  int ci = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float f = g.occ[pp + (c & 1) * s0 + ((c >> 1) & 1) * s1 + ((c >> 2) & 1)];
    ci |= (f > g.iso) << c;
  }
  return ci;
}

// triangles of the cell whose low corner is grid point p
struct TriCount {
  McGrid g;
  __device__ __forceinline__ int operator()(long long pp) const {
    const long long p = pp % g.npts();
    const int i2 = (int)(p % g.Q), i1 = (int)((p / g.Q) % g.Q), i0 = (int)(p / ((long long)g.Q * g.Q));
    if (i0 + 1 >= g.Q || i1 + 1 >= g.Q || i2 + 1 >= g.Q) return 0;
    return MC_NTRI[mc_cube_index(g, pp)];
  }
};

constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_BLK = SCAN_T * SCAN_E;

__device__ __forceinline__ int block_sum(int v, int* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  __syncthreads();
  return t;
}

template <class F>
__global__ __launch_bounds__(SCAN_T) void scan_sums_kernel(F f, long long n, int* bsum) {
  __shared__ int sh[SCAN_T / 64];
  const long long base = (long long)blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_E;
  int s = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) if (base + e < n) s += f(base + e);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}

// single block: exclusive scan of nblk block sums in place, total -> bsum[nblk]
__global__ __launch_bounds__(1024) void scan_bsums_kernel(int* bsum, int nblk) {
  __shared__ int sh[1024];
  int carry = 0;
  for (int c0 = 0; c0 < nblk; c0 += 1024) {
    const int i = c0 + threadIdx.x;
    const int v = i < nblk ? bsum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) bsum[i] = carry + sh[threadIdx.x] - v;
    carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[nblk] = carry;
}

// exclusive scan written to out; offsets[k] = out[k * per_shape], offsets[nshape] = total
template <class F>
__global__ __launch_bounds__(SCAN_T) void scan_apply_kernel(F f, long long n, const int* bsum, int* out, long long per_shape,
                                                            int nshape, int* offsets, int nblk) {
  __shared__ int sh[SCAN_T];
  const long long base = (long long)blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_E;
  int v[SCAN_E], s = 0;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) { v[e] = base + e < n ? f(base + e) : 0; s += v[e]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < SCAN_T; o <<= 1) {
    const int t = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int e = 0; e < SCAN_E; ++e) {
    const long long i = base + e;
    if (i < n) {
      out[i] = run;
      if (i % per_shape == 0) offsets[i / per_shape] = run;
    }
    run += v[e];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[nshape] = bsum[nblk];
}

struct McBox { float lo[3], hi[3]; };

__global__ __launch_bounds__(256) void mc_verts_kernel(McGrid g, const int* __restrict__ vid, const int* __restrict__ voff,
                                                       McBox box, float* __restrict__ verts) {
  const long long n3 = g.npts();
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)g.B * n3 * 3) return;
  const int a = (int)(i % 3);
  const long long pp = i / 3, p = pp % n3;
  int idx[3] = {(int)(p / ((long long)g.Q * g.Q)), (int)((p / g.Q) % g.Q), (int)(p % g.Q)};
  if (idx[a] + 1 >= g.Q) return;
  const long long step = a == 0 ? (long long)g.Q * g.Q : (a == 1 ? g.Q : 1);
  const float f0 = g.occ[pp], f1 = g.occ[pp + step];
  if ((f0 > g.iso) == (f1 > g.iso)) return;
  const float t = __fdiv_rn(g.iso - f0, f1 - f0);
  const int b = (int)(pp / n3);
  float* o = verts + 3ll * vid[i];   // global vertex id (shape b's vertices start at voff[b])
  (void)voff; (void)b;
  const float inv = (float)(g.Q - 1);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float pos = (float)idx[d] + (d == a ? t : 0.f);
    o[d] = __fmaf_rn(__fdiv_rn(pos, inv), box.hi[d] - box.lo[d], box.lo[d]);
  }
}

__global__ __launch_bounds__(256) void mc_faces_kernel(McGrid g, const int* __restrict__ vid, const int* __restrict__ toff,
                                                       const int* __restrict__ voff, int* __restrict__ faces) {
  const long long n3 = g.npts();
  const long long pp = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pp >= (long long)g.B * n3) return;
  const long long p = pp % n3;
  const int i2 = (int)(p % g.Q), i1 = (int)((p / g.Q) % g.Q), i0 = (int)(p / ((long long)g.Q * g.Q));
  if (i0 + 1 >= g.Q || i1 + 1 >= g.Q || i2 + 1 >= g.Q) return;
  const int ci = mc_cube_index(g, pp);
  const int nt = MC_NTRI[ci];
  if (!nt) return;
  const int b = (int)(pp / n3);
  const int vbase = voff[b];
  const long long s0 = (long long)g.Q * g.Q, s1 = g.Q;
  int* o = faces + 3ll * toff[pp];
  for (int k = 0; k < 3 * nt; ++k) {
    const int e = MC_TRI[ci][k];
    const int a = e >> 2, u = e & 1, v = (e >> 1) & 1;
    // edge e runs along axis a from the corner whose other two offsets (increasing axis order) are (u, v)
    const int d0 = a == 0 ? 0 : u, d1 = a == 0 ? u : (a == 1 ? 0 : v), d2 = a == 2 ? 0 : v;
    const long long q = pp + d0 * s0 + d1 * s1 + d2;
    o[k] = vid[3 * q + a] - vbase;
  }
}

template <class F>
int run_scan(F f, long long n, int* bsum, int* out, long long per_shape, int nshape, int* offsets, hipStream_t st) {
  const int nblk = (int)((n + SCAN_BLK - 1) / SCAN_BLK);
  hipLaunchKernelGGL((scan_sums_kernel<F>), dim3(nblk), dim3(SCAN_T), 0, st, f, n, bsum);
  hipLaunchKernelGGL(scan_bsums_kernel, dim3(1), dim3(1024), 0, st, bsum, nblk);
  hipLaunchKernelGGL((scan_apply_kernel<F>), dim3(nblk), dim3(SCAN_T), 0, st, f, n, bsum, out, per_shape, nshape, offsets, nblk);
  return nblk;
}

inline size_t al(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace

extern "C" {

// workspace: vid int[B*3*Q^3] | toff int[B*Q^3] | block sums
size_t sfmi_mc_workspace_bytes(int B, int Q) {
  const size_t n3 = (size_t)Q * Q * Q * B;
  const size_t nb = (3 * n3 + SCAN_BLK - 1) / SCAN_BLK + 2;
  return al(3 * n3 * 4) + al(n3 * 4) + al(nb * 4) * 2;
}

// pass 1: offsets (device, 2*(B+1) ints) = vertex offsets [B+1] then triangle offsets [B+1] (exclusive, per shape)
int sfmi_mc_count_f32(const float* occ, float iso, int B, int Q, void* workspace, int* offsets, void* stream) {
  if (!occ || !workspace || !offsets || B <= 0 || Q < 2 || (long long)B * Q * Q * Q * 3 >= (1ll << 31)) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t n3 = (size_t)Q * Q * Q * B;
  char* w = (char*)workspace;
  int* vid = (int*)w;
  int* toff = (int*)(w + al(3 * n3 * 4));
  const size_t nb = (3 * n3 + SCAN_BLK - 1) / SCAN_BLK + 2;
  int* bs0 = (int*)(w + al(3 * n3 * 4) + al(n3 * 4));
  int* bs1 = (int*)((char*)bs0 + al(nb * 4));
  McGrid g{occ, iso, B, Q};
  run_scan(EdgeCount{g}, (long long)3 * n3, bs0, vid, (long long)3 * Q * Q * Q, B, offsets, st);
  run_scan(TriCount{g}, (long long)n3, bs1, toff, (long long)Q * Q * Q, B, offsets + B + 1, st);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// pass 2: verts (V,3) f32 in the box [lo,hi]^3 mapped as array2mesh does, faces (T,3) int32 LOCAL to each shape
// (shape b: vertices offsets[b]..offsets[b+1], triangles offsets[B+1+b]..offsets[B+2+b]).
int sfmi_mc_emit_f32(const float* occ, float iso, int B, int Q, const void* workspace, const int* offsets, float lo0,
                     float lo1, float lo2, float hi0, float hi1, float hi2, float* verts, int* faces, void* stream) {
  if (!occ || !workspace || !offsets || !verts || !faces || B <= 0 || Q < 2) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t n3 = (size_t)Q * Q * Q * B;
  const char* w = (const char*)workspace;
  const int* vid = (const int*)w;
  const int* toff = (const int*)(w + al(3 * n3 * 4));
  McGrid g{occ, iso, B, Q};
  McBox box{{lo0, lo1, lo2}, {hi0, hi1, hi2}};
  hipLaunchKernelGGL(mc_verts_kernel, dim3((unsigned)((3 * n3 + 255) / 256)), dim3(256), 0, st, g, vid, offsets, box, verts);
  hipLaunchKernelGGL(mc_faces_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, g, vid, toff, offsets, faces);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
