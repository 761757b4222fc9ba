// Fused implicit-decoder SDF/occupancy query for gfx950 (MI355X).
//
// Replaces LocalDecoder.sample_grid_feature + the 5-block conditioned MLP
// (reference shapeformer/models/vqdif/dec.py:62-100, layers.py:39-48) — ~40
// separate ATen launches that each round-trip a (B,N,32) f32 tensor — with ONE
// kernel: per 32-point tile one wave does the trilinear 8-corner gather from a
// channels-last (B,G,G,G,32) grid and then runs all 16 matrix layers as a
// dependent chain of v_mfma_f32_32x32x2_f32 (exact f32) with activations kept
// in the MFMA C/D register layout end to end:
//
//   D[co][pt] = sum_k W[co][k] * X[k][pt]   (A = weights, B = activations)
//
// C/D layout of 32x32x2: lane l holds column pt = l&31 and rows
// co(t,hi) = (t&3) + 8*(t>>2) + 4*hi for t = 0..15, hi = l>>5.  The B operand of
// instruction t wants X[k(t,hi)][pt] from lane (pt,hi) — choosing the K order
// k(t,hi) = co(t,hi) makes register t of the previous layer's output *be* the B
// operand of instruction t of the next layer: no cross-lane movement, no LDS
// round trip for activations.  Weights are staged once per workgroup in LDS,
// pre-permuted on the host so that lane l reads its four A values for
// t = 4g..4g+3 with one conflict-free ds_read_b128.
//
// Roofline: 31 488 FLOP/pt vs 16 B/pt -> MFMA(f32)-bound.
#include "sfmi_common.h"

#define SDF_NL 15                 // fc_c[i], fc_0[i], fc_1[i] for i = 0..4
#define SDF_OFF_W 0               // [15][4][64][4]
#define SDF_OFF_FCP 15360         // [2][64]
#define SDF_OFF_BC 15488          // [5][32]  bc_i + b1_{i-1}
#define SDF_OFF_B0 15648          // [5][32]
#define SDF_OFF_B1L 15808         // [32]     b1_4
#define SDF_OFF_WOUT 15840        // [32]
#define SDF_OFF_BOUT 15872        // [1] (+3 pad)
#define SDF_PACK_FLOATS 15876

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.0f); }

struct SdfAxis {
  int i0, i1;
  float w0, w1;
};

// grid_sample(align_corners=True, padding_mode='border') un-normalisation of one axis
// (ATen grid_sampler_unnormalize + clip_coordinates), G = grid size.
__device__ __forceinline__ SdfAxis sdf_axis(float x_api, int G) {
  float u = sfmi_normalize(x_api * 0.5f);       // vqdif.py:71 Xtg/2 ; dec.py:63
  float v = 2.0f * u - 1.0f;                    // dec.py:65
  float ix = ((v + 1.0f) / 2.0f) * (float)(G - 1);
  ix = fminf((float)(G - 1), fmaxf(ix, 0.0f));
  float f0 = floorf(ix);
  SdfAxis a;
  a.i0 = (int)f0;
  a.i1 = min(a.i0 + 1, G - 1);
  a.w1 = ix - f0;
  a.w0 = (f0 + 1.0f) - ix;
  return a;
}

// AFF: the decoder grid's last GroupNorm (updown.py:119-132: ... Conv, ReLU, GroupNorm of the Upsampler's final block) is handed over as
// its per-(shape, channel) affine instead of being applied to the 64^3 x 32 grid in a pass of its own (67 MB of traffic per shape): the
// trilinear weights of the 'border' gather sum to one, so  interp(x * scale + shift) = interp(x) * scale + shift  - 16 FMAs per lane on
// the gathered features (fp32 rounding differs from the applied form by ~1e-7 of the feature).
template <bool GRID_MODE, bool AFF = false>
__global__ __launch_bounds__(512, 4) void sdf_query_kernel(
    const float* __restrict__ xyz,      // (B,N,3) in [-1,1]           (!GRID_MODE)
    const float* __restrict__ axis,     // (Q) f32 axis table          (GRID_MODE: pt = (ix*Q+iy)*Q+iz)
    const float* __restrict__ axis_x,   // the table the SLOWEST lattice index reads: axis itself, or axis + x0 for the slab of planes x0 .. (N / Q^2 of them)
    const float* __restrict__ grid,     // (B,G,G,G,32) channels-last, [z][y][x][c]
    const float* __restrict__ wpack,    // SDF_PACK_FLOATS
    float* __restrict__ out,            // (B,N)
    int B, long long N, int G, int Q, int apply_sigmoid,
    const float* __restrict__ aff_scale = nullptr, const float* __restrict__ aff_shift = nullptr) {   // AFF: (B,32) each
  extern __shared__ __attribute__((aligned(16))) float lds[];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(wpack);
    f32x4* dst = reinterpret_cast<f32x4*>(lds);
    for (int i = threadIdx.x; i < SDF_PACK_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5;
  const int pl = lane & 31;
  // tile indices are 32-bit (the host entries bound B * ceil(N/32) < 2^31): no 64-bit division in the persistent loop
  const unsigned tiles_per_shape = (unsigned)((N + 31) >> 5);
  const unsigned total_tiles = tiles_per_shape * (unsigned)B;
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
  const unsigned wave_gid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);

  const f32x4* ldsW = reinterpret_cast<const f32x4*>(lds + SDF_OFF_W);
  const float wp0 = lds[SDF_OFF_FCP + lane];
  const float wp1 = lds[SDF_OFF_FCP + 64 + lane];

  for (unsigned tile = wave_gid; tile < total_tiles; tile += nwaves) {
    const int b = (int)(tile / tiles_per_shape);
    long long pt = (long long)(tile - (unsigned)b * tiles_per_shape) * 32 + pl;
    const bool valid = pt < N;
    if (!valid) pt = N - 1;

    float px, py, pz;
    if (GRID_MODE) {
      // nputil.makeGrid 'ij' flatten: x slowest, z fastest (xgutils/nputil.py:618-654)
      // 32-bit index arithmetic (the host entry bounds Q^3 < 2^31): the 64-bit divisions cost the registers that used to spill
      const unsigned p32 = (unsigned)pt, uq = (unsigned)Q;
      const unsigned r = p32 / uq;
      const int iz = (int)(p32 - r * uq);
      const int ixx = (int)(r / uq);
      const int iy = (int)(r - (unsigned)ixx * uq);
      px = axis_x[ixx]; py = axis[iy]; pz = axis[iz];
    } else {
      const float* p = xyz + ((long long)b * N + pt) * 3;
      px = p[0]; py = p[1]; pz = p[2];
    }
    const SdfAxis ax = sdf_axis(px, G), ay = sdf_axis(py, G), az = sdf_axis(pz, G);

    // ---- trilinear gather: lane (pt,hi) gets channels j + 8g + 4hi ----------
    f32x16 c;
#pragma unroll
    for (int t = 0; t < 16; ++t) c[t] = 0.0f;
    const float* gb = grid + (long long)b * G * G * G * 32 + 4 * hi;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      // ATen order: tnw,tne,tsw,tse,bnw,bne,bsw,bse  (t/b = z0/z1, n/s = y0/y1, w/e = x0/x1)
      const int dz = corner >> 2, dy = (corner >> 1) & 1, dx = corner & 1;
      const int zi = dz ? az.i1 : az.i0, yi = dy ? ay.i1 : ay.i0, xi = dx ? ax.i1 : ax.i0;
      const float w = ((dx ? ax.w1 : ax.w0) * (dy ? ay.w1 : ay.w0)) * (dz ? az.w1 : az.w0);
      const f32x4* cp = reinterpret_cast<const f32x4*>(gb + (((long long)zi * G + yi) * G + xi) * 32);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = cp[2 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[4 * g + j] = fmaf(v[j], w, c[4 * g + j]);
      }
    }

    if (AFF) {
      const f32x4* sp = reinterpret_cast<const f32x4*>(aff_scale + b * 32 + 4 * hi);
      const f32x4* tp = reinterpret_cast<const f32x4*>(aff_shift + b * 32 + 4 * hi);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sc = sp[2 * g], sh = tp[2 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[4 * g + j] = fmaf(c[4 * g + j], sc[j], sh[j]);
      }
    }

    // ---- fc_p (K = 4: x,y,z,1 -> bias folded) ------------------------------
    const float hx = px * 0.5f, hy = py * 0.5f, hz = pz * 0.5f;  // dec.py:88 p = Xtg/2
    f32x16 net;
#pragma unroll
    for (int t = 0; t < 16; ++t) net[t] = 0.0f;
    net = MFMA(wp0, hi ? hy : hx, net);
    net = MFMA(wp1, hi ? 1.0f : hz, net);

#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
      const f32x4* Wc = ldsW + (3 * i + 0) * 256 + lane;
      const f32x4* W0 = ldsW + (3 * i + 1) * 256 + lane;
      const f32x4* W1 = ldsW + (3 * i + 2) * 256 + lane;
      const f32x4* bc = reinterpret_cast<const f32x4*>(lds + SDF_OFF_BC + 32 * i + 4 * hi);
      const f32x4* b0 = reinterpret_cast<const f32x4*>(lds + SDF_OFF_B0 + 32 * i + 4 * hi);
      // net += Wc_i c + (bc_i + b1_{i-1})
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w = Wc[64 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) net = MFMA(w[j], c[4 * g + j], net);
      }
      f32x16 h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bb = bc[2 * g], b00 = b0[2 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          net[4 * g + j] += bb[j];
          h[4 * g + j] = b00[j];
        }
      }
      // h = W0 relu(net) + b0
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w = W0[64 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) h = MFMA(w[j], relu(net[4 * g + j]), h);
      }
      // net += W1 relu(h)      (b1_i is folded into the next bc / the tail)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 w = W1[64 * g];
#pragma unroll
        for (int j = 0; j < 4; ++j) net = MFMA(w[j], relu(h[4 * g + j]), net);
      }
    }

    // ---- fc_out(relu(net + b1_4)) -------------------------------------------
    const f32x4* b1l = reinterpret_cast<const f32x4*>(lds + SDF_OFF_B1L + 4 * hi);
    const f32x4* wo = reinterpret_cast<const f32x4*>(lds + SDF_OFF_WOUT + 4 * hi);
    float r = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 bb = b1l[2 * g], ww = wo[2 * g];
#pragma unroll
      for (int j = 0; j < 4; ++j) r = fmaf(ww[j], relu(net[4 * g + j] + bb[j]), r);
    }
    r += __shfl_xor(r, 32, 64);
    r += lds[SDF_OFF_BOUT];
    if (apply_sigmoid) r = 1.0f / (1.0f + __expf(-r));
    if (valid && hi == 0) out[(long long)b * N + pt] = r;
  }
}

extern "C" {

size_t sfmi_sdf_pack_floats(void) { return SDF_PACK_FLOATS; }

// Host-side packer: reference LocalDecoder tensors (row-major, torch Linear layout) ->
// the kernel's fragment-ordered buffer.  fc_c_w/b, fc0_w/b, fc1_w/b: 5 consecutive
// (32,32)/(32) tensors each.
int sfmi_sdf_pack_weights(const float* fc_p_w /*32x3*/, const float* fc_p_b, const float* fc_c_w,
                          const float* fc_c_b, const float* fc0_w, const float* fc0_b,
                          const float* fc1_w, const float* fc1_b, const float* fc_out_w /*32*/,
                          const float* fc_out_b /*1*/, float* out /*SDF_PACK_FLOATS*/) {
  if (!fc_p_w || !out) return SFMI_EINVAL;
  for (int i = 0; i < 5; ++i) {
    const float* Ws[3] = {fc_c_w + i * 1024, fc0_w + i * 1024, fc1_w + i * 1024};
    for (int m = 0; m < 3; ++m)
      for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 4; ++j) {
            int co = l & 31, k = 8 * g + 4 * (l >> 5) + j;
            out[SDF_OFF_W + (((3 * i + m) * 4 + g) * 64 + l) * 4 + j] = Ws[m][co * 32 + k];
          }
    for (int ch = 0; ch < 32; ++ch) {
      out[SDF_OFF_BC + 32 * i + ch] = fc_c_b[32 * i + ch] + (i > 0 ? fc1_b[32 * (i - 1) + ch] : 0.0f);
      out[SDF_OFF_B0 + 32 * i + ch] = fc0_b[32 * i + ch];
    }
  }
  for (int l = 0; l < 64; ++l) {
    int co = l & 31, hi = l >> 5;
    out[SDF_OFF_FCP + l] = fc_p_w[co * 3 + (hi ? 1 : 0)];
    out[SDF_OFF_FCP + 64 + l] = hi ? fc_p_b[co] : fc_p_w[co * 3 + 2];
  }
  for (int ch = 0; ch < 32; ++ch) {
    out[SDF_OFF_B1L + ch] = fc1_b[32 * 4 + ch];
    out[SDF_OFF_WOUT + ch] = fc_out_w[ch];
  }
  out[SDF_OFF_BOUT] = fc_out_b[0];
  out[SDF_OFF_BOUT + 1] = out[SDF_OFF_BOUT + 2] = out[SDF_OFF_BOUT + 3] = 0.0f;
  return SFMI_OK;
}

// nputil.sigmoid over a logits tensor (the inference drivers keep the reference's `logits` result key and extract the mesh from the
// occupancy, shapeformer.py:382-391 / vqdif.py:243-269): the expression of the fused epilogue above, as a grid-stride pass
__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  const long long n4 = n >> 2, stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + __expf(-v[e]));
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = 1.0f / (1.0f + __expf(-x[i]));
}

static int sdf_grid_dim(long long total_tiles) {
  long long wgs = (total_tiles + 7) / 8;
  if (wgs > g_sfmi_tune.sdf_blocks) wgs = g_sfmi_tune.sdf_blocks;   // default 512 = 256 CUs x 2 resident workgroups (LDS-limited), persistent tile loop
  if (wgs < 1) wgs = 1;
  return (int)wgs;
}

// replaces LocalDecoder.sample_grid_feature + MLP: dec.py:62-100 (arbitrary query points)
int sfmi_sdf_query_f32(const float* xyz, const float* grid_cl, const float* wpack, float* out, int B,
                       long long N, int G, int apply_sigmoid, void* stream) {
  if (!xyz || !grid_cl || !wpack || !out || B <= 0 || N <= 0 || G < 2) return SFMI_EINVAL;
  long long tiles = ((N + 31) >> 5) * B;
  if (tiles >= (1ll << 31)) return SFMI_EINVAL;
  hipLaunchKernelGGL(sdf_query_kernel<false>, dim3(sdf_grid_dim(tiles)), dim3(512),
                     SDF_PACK_FLOATS * sizeof(float), (hipStream_t)stream, xyz, nullptr, nullptr, grid_cl, wpack,
                     out, B, N, G, 0, apply_sigmoid);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// structured Q^3 query grid (nputil.makeGrid 'ij', shapeformer.py:219-220 / vqdif.py:223-224):
// coordinates synthesised from the Q-entry axis table -> 4 B/pt of HBM traffic.
int sfmi_sdf_query_grid_slab_f32(const float* axis, int Q, int x0, int x1, const float* grid_cl, const float* wpack, float* out,
                                 int B, int G, int apply_sigmoid, void* stream);
int sfmi_sdf_query_grid_f32(const float* axis, int Q, const float* grid_cl, const float* wpack, float* out,
                            int B, int G, int apply_sigmoid, void* stream) {
  return sfmi_sdf_query_grid_slab_f32(axis, Q, 0, Q, grid_cl, wpack, out, B, G, apply_sigmoid, stream);
}
// the planes x0 <= ix < x1 of the same lattice (ix is the slowest index of the 'ij' flatten, so a slab is a contiguous range of lattice
// points): out (B, (x1 - x0) Q^2).  Every point's arithmetic is that of the whole-lattice call - the slabs of a lattice, computed by
// different processes, concatenate to its result bit for bit (the z-slab split of ONE shape, dist.sdf_query_sharded).
int sfmi_sdf_query_grid_aff_f32(const float* axis, int Q, int x0, int x1, const float* grid_cl, const float* aff_scale, const float* aff_shift,
                                const float* wpack, float* out, int B, int G, int apply_sigmoid, void* stream);
int sfmi_sdf_query_grid_slab_f32(const float* axis, int Q, int x0, int x1, const float* grid_cl, const float* wpack, float* out,
                                 int B, int G, int apply_sigmoid, void* stream) {
  return sfmi_sdf_query_grid_aff_f32(axis, Q, x0, x1, grid_cl, nullptr, nullptr, wpack, out, B, G, apply_sigmoid, stream);
}
// the same on a feature grid whose last GroupNorm has NOT been applied: aff_scale / aff_shift (B,32) = that GroupNorm's per-(shape, channel)
// affine (sfmi_groupnorm_coeffs_f32), applied to the interpolated features inside the kernel (both NULL: the grid is final).
// Replaces the apply pass of updown.py:119-132's last GroupNorm in front of dec.py:62-100.
int sfmi_sdf_query_grid_aff_f32(const float* axis, int Q, int x0, int x1, const float* grid_cl, const float* aff_scale, const float* aff_shift,
                                const float* wpack, float* out, int B, int G, int apply_sigmoid, void* stream) {
  if ((aff_scale == nullptr) != (aff_shift == nullptr)) return SFMI_EINVAL;
  if (!axis || !grid_cl || !wpack || !out || B <= 0 || Q <= 0 || Q > 1024 || G < 2 || x0 < 0 || x1 > Q || x0 >= x1) return SFMI_EINVAL;   // Q^3 < 2^31 (32-bit lattice index)
  long long N = (long long)(x1 - x0) * Q * Q;
  long long tiles = ((N + 31) >> 5) * B;
  if (tiles >= (1ll << 31)) return SFMI_EINVAL;
  if (aff_scale)
    hipLaunchKernelGGL((sdf_query_kernel<true, true>), dim3(sdf_grid_dim(tiles)), dim3(512),
                       SDF_PACK_FLOATS * sizeof(float), (hipStream_t)stream, nullptr, axis, axis + x0, grid_cl, wpack,
                       out, B, N, G, Q, apply_sigmoid, aff_scale, aff_shift);
  else
    hipLaunchKernelGGL((sdf_query_kernel<true, false>), dim3(sdf_grid_dim(tiles)), dim3(512),
                       SDF_PACK_FLOATS * sizeof(float), (hipStream_t)stream, nullptr, axis, axis + x0, grid_cl, wpack,
                       out, B, N, G, Q, apply_sigmoid, nullptr, nullptr);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// replaces nputil.sigmoid(logits) of the inference drivers (vqdif.py:262, shapeformer.py:388); x and y 16-byte aligned, may alias
int sfmi_sigmoid_f32(const float* x, float* y, long long n, void* stream) {
  if (!x || !y || n <= 0) return SFMI_EINVAL;
  const long long wgs = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(sigmoid_kernel, dim3((unsigned)(wgs < 1 ? 1 : wgs > 4096 ? 4096 : wgs)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
