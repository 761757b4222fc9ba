// Channels-last 3-D convolution as an implicit GEMM on f32 MFMA (gfx950), plus the GroupNorm /
// pooling / concat helpers around it.
//
// Replaces the cuDNN conv3d + ATen group_norm/relu/max_pool3d/interpolate/cat launches of
// Downsampler (updown.py:101-118), UNet3D (unet3d.py:79-144,195-293,449-474) and Upsampler
// (updown.py:119-132).  Layout is (B,D,H,W,C) so that a voxel's channels are one contiguous line —
// the layout the SDF-query kernel gathers from.  Fusions done here instead of separate passes:
//   * GroupNorm APPLY of the producer is folded into this conv's input load (x*scale[b,c]+shift[b,c],
//     zero padding applied after the affine, as the reference pads the normalised tensor);
//   * nearest x2 upsampling is an address shift (>>1) on the input coordinate, never materialised;
//   * bias / ReLU in the epilogue.
// GEMM view: D[co][voxel] = sum_{tap,cin} W[tap][co][cin] * X[voxel+tap][cin]; A = weights, B = activations,
// 32x32x2 f32 MFMA (exact f32).  Workgroup = 4 waves, LDS double-buffered K-chunks of 16 input channels, two register sets
// (the global loads of chunk c+2 are in flight during the MFMAs of chunk c), one barrier per chunk; the tap walk and the bounds
// tests are wave-uniform scalars + per-voxel bit masks, the GroupNorm affine is applied when a register set is written to LDS.
#include "sfmi_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define KC 16
#define LDS_STRIDE 20  // floats per row: 16 + 4 pad -> conflict-free ds_read_b128 over 16 consecutive rows

struct ConvArgs {
  const float* x; const float* wT; const float* in_scale; const float* in_shift; const float* bias; float* y;
  const float* resid;                     // optional residual added in the epilogue (same row mapping as y)
  long long out_group, out_group_stride;  // output row remap: m -> (m/out_group)*out_group_stride + m%out_group
  int B, Di, Hi, Wi, Do, Ho, Wo, Cin, Cout, KS, stride, pad, up, relu /*0 none, 1 relu, 2 gelu(erf)*/;
  // sub-pixel mode (sp_on): this launch computes the outputs of ONE parity (spz,spy,spx) of a nearest-x2-upsampled conv:
  // leading pad per axis padz/pady/padx (instead of `pad`), output voxel (z,y,x) of the (Do,Ho,Wo) lattice is written to
  // voxel (2z+spz, 2y+spy, 2x+spx) of the (2Do,2Ho,2Wo) result
  int sp_on, spz, spy, spx, padz, pady, padx;
  // ST instances: per-(tile, output channel) sum / sum of squares of the written values (f64 pairs, the layout gn_coeffs_kernel reads:
  // [shape][stats_S splits][Cout][2]); this launch's tiles of a shape are splits stats_sp0 .. (a sub-pixel parity launch has its own range)
  double* stats; int stats_S, stats_sp0;
};

// XR ("x reuse", stride-1 convolutions whose M tile is made of whole x-rows of the output; narrow output-channel tiles): the taps of
// one (dz, dy) pair read the SAME input voxels shifted by one in x, so a chunk stages the tile's input rows ONCE - each x-row
// framed by its zero-padding columns - and multiplies them KS times at LDS row offsets 0 .. KS-1 against KS weight tiles.  One third
// (one half for the 2^3 sub-pixel convolutions) of the global loads, LDS writes and barriers per MFMA of the plain form, which
// re-stages the activation tile for every tap; with 32 or 64 output channels per workgroup that staging is what bounds the kernel.
// J = 32-voxel tiles per wave (2; 4 for the 32-channel x-reuse tile of 512 voxels: with 32 output channels a 256-voxel tile gives a wave
// only 48 MFMAs between barriers and a workgroup 18 chunks over which to spread its voxel decode, first loads and epilogue).
// ST: the epilogue also reduces the tile's outputs to per-channel (sum, sum of squares) partials - the GroupNorm statistics of the NEXT layer
// (updown.py:119-132: Conv, ReLU, GroupNorm) without a pass of their own over the output (WM = 1 instances: a wave holds all N_T channels).
template <int CO_TILES, int WM, int WN, bool UPS, bool XR = false, int J = 2, bool ST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CO_TILES == 1 && !XR ? 3 : 2))) void conv3d_igemm_kernel(ConvArgs a) {
  constexpr int N_T = 32 * CO_TILES * WM;
  constexpr int M_T = 32 * J * WN;
  constexpr int AP = M_T / 64;                   // activation float4 rows per thread (64 rows per pass of the 256 threads)
  constexpr int NS = J == 4 ? 1 : 2;             // register sets of the global loads (J = 4: a chunk is long enough for one)
  constexpr int W_ROWS = (N_T * 4 + 255) / 256;  // weight float4 rows per thread
  constexpr int XT = XR ? 3 : 1;                 // weight tiles per chunk (taps along x; KS <= 3)
  // 128 output channels with x reuse: three 128-row weight tiles per buffer.  Rows of 16 floats (no padding column; the 16-byte
  // segment of a row is XOR-ed with bits 2-3 of the row index, so 16 consecutive rows still cover all 64 banks) keep the
  // workgroup under 80 KB and two of them resident per CU.
  constexpr bool SWZ = XR && (N_T == 128 || J == 4);
  constexpr int LS = SWZ ? 16 : LDS_STRIDE;
  static_assert(!(XR && UPS), "x reuse needs the direct addressing");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // XR: an x-row of Wo voxels occupies Wo + KS - 1 LDS rows: padL zero columns, the Wo real ones, padR zero columns
  const int xr_rw = XR ? a.Wo + a.KS - 1 : 0;
  const int xr_padl = XR ? (a.sp_on ? a.padx : a.pad) : 0;
  const int AROWS = XR ? (SWZ ? ((M_T / a.Wo) * xr_rw + 15) & ~15 : (M_T / a.Wo) * xr_rw) : M_T;   // activation rows per LDS buffer
  float* act_lds = lds;                     // [2][AROWS][LS]   (SWZ: AROWS and N_T are multiples of 16, the swizzle needs only
  float* wgt_lds = lds + 2 * AROWS * LS;    // [2][XT][N_T][LS]  the row index inside a buffer)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, pl = lane & 31;
  const int wm = wave / WN, wn = wave % WN;
  const int srow = tid >> 2, seg = tid & 3;
  const long long M = (long long)a.B * a.Do * a.Ho * a.Wo;
  // XCD-aware block order: block b runs on XCD b%8 (observed, speed only).  Logical ids are laid out so that the
  // blocks that share an activation (M) tile - they differ only in the output-channel tile - are consecutive ON ONE
  // XCD and hit that XCD's L2 instead of re-fetching the tile from HBM once per channel tile (PMC: 8x over-fetch).
  const int nbn = a.Cout / N_T;
  const long long nb = (long long)gridDim.x;
  long long lid = blockIdx.x;
  {
    const long long q = nb / 8, r = nb % 8, xcd = lid % 8, k = lid / 8;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;   // bijective for any nb
  }
  const long long m0 = (lid / nbn) * M_T;
  const int n0 = (int)(lid % nbn) * N_T;
  const int Dv = a.Di << a.up, Hv = a.Hi << a.up, Wv = a.Wi << a.up;

  // decode the output voxels this thread stages
  int vb[AP], vz[AP], vy[AP], vx[AP];
  bool vok[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    long long m = m0 + srow + 64 * i;
    vok[i] = m < M;
    if (!vok[i]) m = M - 1;
    int xw = (int)(m % a.Wo); long long r = m / a.Wo;
    int yh = (int)(r % a.Ho); r /= a.Ho;
    int zd = (int)(r % a.Do);
    vb[i] = (int)(r / a.Do);
    vz[i] = zd * a.stride - (a.sp_on ? a.padz : a.pad); vy[i] = yh * a.stride - (a.sp_on ? a.pady : a.pad);
    vx[i] = xw * a.stride - (a.sp_on ? a.padx : a.pad);
  }
  const int cpt = a.Cin / KC;                 // chunks per tap
  const int nchunks = (XR ? a.KS * a.KS : a.KS * a.KS * a.KS) * cpt;   // XR: one chunk serves the KS taps along x

  // Per-chunk work kept off the vector unit (with one 32-column tile per wave the MFMAs of a chunk take ~1000 cycles, and
  // the old per-chunk tap decode + bounds tests + 64-bit address chains of 4 voxels took about as long): per voxel ONE base
  // offset and three 3-bit per-axis validity masks are computed here; inside the loop the tap walks (dz,dy,dx,chunk) as
  // wave-uniform scalars and a voxel costs a shift-and-test plus one 64-bit add.  UPS (nearest-x2 folded into the address,
  // only the direct form of an up-sampling conv) keeps the general arithmetic.
  long long vbase[AP];
  int vmask[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int mk = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d < a.KS && vz[i] + d >= 0 && vz[i] + d < Dv) mk |= 1 << d;
      if (d < a.KS && vy[i] + d >= 0 && vy[i] + d < Hv) mk |= 8 << d;
      if (d < a.KS && vx[i] + d >= 0 && vx[i] + d < Wv) mk |= 64 << d;
    }
    vmask[i] = vok[i] ? mk : 0;
    vbase[i] = ((((long long)vb[i] * a.Di + vz[i]) * a.Hi + vy[i]) * a.Wi + vx[i]) * a.Cin + seg * 4;
  }
  const bool one_b = (vb[0] == vb[AP - 1]);   // the tile lies inside one shape: its GroupNorm affine is loaded once per chunk
  long long wbase[W_ROWS];
#pragma unroll
  for (int i = 0; i < W_ROWS; ++i) wbase[i] = (long long)(n0 + srow + 64 * i) * a.Cin + seg * 4;
  int t_dz = 0, t_dy = 0, t_dx = 0, t_cc = 0, t_tap = 0;   // wave-uniform walk over (tap, channel chunk)

  // LDS row of the voxel this thread stages / of the voxels this lane multiplies (XR: inside its framed x-row)
  int srow_l[AP], mrow_l[J];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int v = srow + 64 * i;
    srow_l[i] = XR ? (v / a.Wo) * xr_rw + xr_padl + v % a.Wo : v;
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int v = wn * 32 * J + j * 32 + pl;
    mrow_l[j] = XR ? (v / a.Wo) * xr_rw + v % a.Wo : v;      // tap dx reads row mrow_l + dx
  }
  if (XR) {   // the frame columns of both buffers are zero for the whole launch (nothing else writes them)
    const int nxr = M_T / a.Wo, fr = a.KS - 1;
    for (int idx = tid; idx < 2 * nxr * fr * (LS / 4); idx += 256) {
      const int q4 = idx % (LS / 4), r = idx / (LS / 4);
      const int f = r % fr, xrow = (r / fr) % nxr, S = r / (fr * nxr);
      const int lrow = xrow * xr_rw + (f < xr_padl ? f : a.Wo + f);
      *reinterpret_cast<f32x4*>(act_lds + (S * AROWS + lrow) * LS + 4 * q4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // column (in floats) of the 16-byte segment a thread stages / a lane reads; the reads of tap d are one row further per tap
  int scol_a[AP], mcol[J == 4 ? 1 : J][XT][2], wcol[2];   // (J = 4 computes the activation column at the read: registers)
  const int scol_w = 4 * (SWZ ? seg ^ ((srow >> 2) & 3) : seg);
#pragma unroll
  for (int i = 0; i < AP; ++i) scol_a[i] = 4 * (SWZ ? seg ^ ((srow_l[i] >> 2) & 3) : seg);
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    wcol[sub] = 4 * (SWZ ? (hi + 2 * sub) ^ ((pl >> 2) & 3) : hi + 2 * sub);
#pragma unroll
    for (int j = 0; j < (J == 4 ? 1 : J); ++j)
#pragma unroll
      for (int d = 0; d < XT; ++d) mcol[j][d][sub] = 4 * (SWZ ? (hi + 2 * sub) ^ (((mrow_l[j] + d) >> 2) & 3) : hi + 2 * sub);
  }

  // two register sets: the global loads of chunk c+2 are issued while chunk c is multiplied (one set gave the loads only the
  // ~1000 MFMA cycles of a single chunk to land).  The GroupNorm affine and the zero padding are applied when a set is
  // written to LDS (x*scale+shift needs the loaded value: done at load time it would stall on the load it was meant to hide).
  f32x4 ra[NS][AP], rw[NS][XT * W_ROWS], rs[NS], rt[NS];
  int rok[NS];
  auto load_chunk = [&](const int S) {
    const int c0 = t_cc * KC + seg * 4;
    rs[S] = f32x4{1.f, 1.f, 1.f, 1.f};
    rt[S] = f32x4{0.f, 0.f, 0.f, 0.f};
    rok[S] = 0;
    if (UPS || (!one_b && J != 4)) {   // (J = 4: the host launches it only where no tile straddles two shapes)  general arithmetic (address-folded up-sampling, tiles that straddle shapes): affine applied here
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int iz = vz[i] + t_dz, iy = vy[i] + t_dy, ix = vx[i] + (XR ? xr_padl : t_dx);   // XR: the voxel's own column
        const bool ok = vok[i] && iz >= 0 && iz < Dv && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const long long off = ((((long long)vb[i] * a.Di + (iz >> a.up)) * a.Hi + (iy >> a.up)) * a.Wi + (ix >> a.up)) * a.Cin + c0;
          v = *reinterpret_cast<const f32x4*>(a.x + off);
          if (a.in_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.in_scale + (long long)vb[i] * a.Cin + c0);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.in_shift + (long long)vb[i] * a.Cin + c0);
            v = v * sc + sh;
          }
        }
        ra[S][i] = v;
        rok[S] |= 1 << i;
      }
    } else {
      const long long toff = (((long long)t_dz * a.Hi + t_dy) * a.Wi + (XR ? xr_padl : t_dx)) * a.Cin + t_cc * KC;
      if (a.in_scale) {
        rs[S] = *reinterpret_cast<const f32x4*>(a.in_scale + (long long)vb[0] * a.Cin + c0);
        rt[S] = *reinterpret_cast<const f32x4*>(a.in_shift + (long long)vb[0] * a.Cin + c0);
      }
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const bool ok = XR ? (((vmask[i] >> t_dz) & (vmask[i] >> (3 + t_dy)) & 1) != 0 && vok[i])     // the own column is always inside
                           : (((vmask[i] >> t_dz) & (vmask[i] >> (3 + t_dy)) & (vmask[i] >> (6 + t_dx)) & 1) != 0);
        ra[S][i] = *reinterpret_cast<const f32x4*>(a.x + (ok ? vbase[i] + toff : (long long)(seg * 4)));   // unconditional load
        rok[S] |= (ok ? 1 : 0) << i;
      }
    }
    const long long woff = (long long)t_tap * a.Cout * a.Cin + t_cc * KC;
#pragma unroll
    for (int d = 0; d < XT; ++d)
#pragma unroll
      for (int i = 0; i < W_ROWS; ++i) {
        const int row = srow + 64 * i;
        if (row < N_T && d < a.KS) rw[S][d * W_ROWS + i] = *reinterpret_cast<const f32x4*>(a.wT + wbase[i] + woff + (long long)d * a.Cout * a.Cin);
      }
    if (++t_cc == cpt) {
      t_cc = 0;
      if (XR) { t_tap += a.KS; if (++t_dy == a.KS) { t_dy = 0; ++t_dz; } }
      else { ++t_tap; if (++t_dx == a.KS) { t_dx = 0; if (++t_dy == a.KS) { t_dy = 0; ++t_dz; } } }
    }
  };
  auto store_chunk = [&](const int R, const int S) {   // register set R -> LDS buffer S
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      f32x4 v = ra[R][i];
      if (a.in_scale) v = v * rs[R] + rt[R];
      if (!((rok[R] >> i) & 1)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(act_lds + ((S * AROWS) + srow_l[i]) * LS + scol_a[i]) = v;
    }
#pragma unroll
    for (int d = 0; d < XT; ++d)
#pragma unroll
      for (int i = 0; i < W_ROWS; ++i) {
        const int row = srow + 64 * i;
        if (row < N_T && d < a.KS) *reinterpret_cast<f32x4*>(wgt_lds + (((S * XT + d) * N_T) + row) * LS + scol_w) = rw[R][d * W_ROWS + i];
      }
  };

  f32x16 acc[CO_TILES][J];
#pragma unroll
  for (int i = 0; i < CO_TILES; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.0f;
  // ACC2 (the 128 x 64 x-reuse tile: UNet3D): blocked accumulation.  An MFMA chain over all K = 27 Cin products (3 456 .. 20 736 in
  // UNet3D) rounds once per instruction and its error grows like sqrt(K): measured 0.9e-6 .. 2.4e-6 of the output's rms per layer
  // against 3.5e-7 for the reference's CPU convolution, and the 10-layer stack with its GroupNorms carried that to 4e-5 rms at the
  // logits - the end-to-end gate at 0.85 .. 1.14 of its width depending on the summation order.  Folding the running tile into a
  // second accumulator every FOLD chunks (384 products) bounds the chain: 32 adds per 384 MFMAs.
  constexpr bool ACC2 = XR && CO_TILES == 1 && WM == 2 && WN == 2 && J == 2;
  constexpr int FOLD = 8;
  f32x16 acc2[ACC2 ? CO_TILES : 1][2];
  int fold_n = 0;
  if (ACC2) {
#pragma unroll
    for (int i = 0; i < CO_TILES; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc2[i][j][t] = 0.0f;
  }
  auto fold = [&]() {
#pragma unroll
    for (int i = 0; i < (ACC2 ? CO_TILES : 0); ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 16; ++t) { acc2[i][j][t] += acc[i][j][t]; acc[i][j][t] = 0.0f; }
  };

  auto multiply = [&](const int S) {
#pragma unroll
    for (int d = 0; d < XT; ++d) {
      if (d >= (XR ? a.KS : 1)) break;
      const float* ab = act_lds + (S * AROWS + d) * LS;
      const float* wb = wgt_lds + ((S * XT + d) * N_T + wm * CO_TILES * 32 + pl) * LS;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        f32x4 bf[J], af[CO_TILES];
#pragma unroll
        for (int j = 0; j < J; ++j) bf[j] = *reinterpret_cast<const f32x4*>(ab + mrow_l[j] * LS + (J == 4 ? 4 * ((hi + 2 * sub) ^ (((mrow_l[j] + d) >> 2) & 3)) : mcol[j][d][sub]));
#pragma unroll
        for (int i = 0; i < CO_TILES; ++i) af[i] = *reinterpret_cast<const f32x4*>(wb + i * 32 * LS + wcol[sub]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int i = 0; i < CO_TILES; ++i)
#pragma unroll
            for (int j = 0; j < J; ++j) acc[i][j] = MFMA(af[i][q], bf[j][q], acc[i][j]);
      }
    }
    if (ACC2 && ++fold_n == FOLD) { fold_n = 0; fold(); }
  };
  load_chunk(0);
  if (NS == 2 && nchunks > 1) load_chunk(NS - 1);
  for (int c = 0; c < nchunks; c += 2) {
    store_chunk(0, 0);
    __syncthreads();
    if (c + NS < nchunks) load_chunk(0);
    multiply(0);
    if (c + 1 < nchunks) {
      store_chunk(NS - 1, 1);
      __syncthreads();
      if (c + 1 + NS < nchunks) load_chunk(NS - 1);
      multiply(1);
    }
  }
  if (ACC2) {   // the last (partial) block, then the epilogue reads `acc`
    fold();
#pragma unroll
    for (int i = 0; i < (ACC2 ? CO_TILES : 0); ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = acc2[i][j];
  }

  // epilogue: lane (voxel, hi) holds couts 8g+4hi+j of each co tile -> 4 float4 stores per tile
  f32x16 ssum[ST ? CO_TILES : 1], ssq[ST ? CO_TILES : 1];
  if (ST) {
#pragma unroll
    for (int i = 0; i < CO_TILES; ++i)
#pragma unroll
      for (int t = 0; t < 16; ++t) { ssum[i][t] = 0.f; ssq[i][t] = 0.f; }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const long long m_in = m0 + wn * 32 * J + j * 32 + pl;
    if (m_in >= M) continue;
    long long m = a.out_group ? (m_in / a.out_group) * a.out_group_stride + m_in % a.out_group : m_in;
    if (a.sp_on) {
      const int xw = (int)(m_in % a.Wo); long long r = m_in / a.Wo;
      const int yh = (int)(r % a.Ho); r /= a.Ho;
      const int zd = (int)(r % a.Do);
      const long long bb = r / a.Do;
      m = ((bb * 2 * a.Do + 2 * zd + a.spz) * 2 * a.Ho + 2 * yh + a.spy) * 2 * a.Wo + 2 * xw + a.spx;
    }
#pragma unroll
    for (int i = 0; i < CO_TILES; ++i) {
      const int cob = n0 + (wm * CO_TILES + i) * 32 + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (a.bias) v = v + *reinterpret_cast<const f32x4*>(a.bias + cob + 8 * g);
        if (a.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (a.relu == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
        }
        if (a.resid) v = v + *reinterpret_cast<const f32x4*>(a.resid + m * a.Cout + cob + 8 * g);
        *reinterpret_cast<f32x4*>(a.y + m * a.Cout + cob + 8 * g) = v;
        if (ST) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { ssum[i][4 * g + e] += v[e]; ssq[i][4 * g + e] = fmaf(v[e], v[e], ssq[i][4 * g + e]); }
        }
      }
    }
  }
  if (ST) {
    // per channel: this lane's voxels (above) -> the 32 voxel lanes of its half-wave (DPP row sums + one cross-row exchange) -> the WN waves
    // (LDS) -> one f64 pair per (tile, channel).  f32 up to here (<= 512 values per channel), f64 across tiles (gn_coeffs_kernel).
    static_assert(!ST || WM == 1, "the statistics epilogue is written for instances whose waves hold all output channels");
    __syncthreads();                       // every wave is done with the operand buffers: LDS is free
    float* red = lds;                      // [WN][CO_TILES][2 (hi)][16][2]
#pragma unroll
    for (int i = 0; i < CO_TILES; ++i)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        float sv = row16_sum(ssum[i][t]), qv = row16_sum(ssq[i][t]);
        sv += __shfl_xor(sv, 16, 64); qv += __shfl_xor(qv, 16, 64);
        if (pl == 0) {
          float* r = red + ((((wn * CO_TILES + i) * 2 + hi) * 16 + t) * 2);
          r[0] = sv; r[1] = qv;
        }
      }
    __syncthreads();
    if (tid < N_T) {
      const int i = tid >> 5, c = tid & 31;              // channel c of co tile i: register t = 4 (c >> 3) + (c & 3) of the lanes with hi = (c >> 2) & 1
      const int t = 4 * (c >> 3) + (c & 3), h = (c >> 2) & 1;
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < WN; ++w) {
        const float* r = red + ((((w * CO_TILES + i) * 2 + h) * 16 + t) * 2);
        sv += (double)r[0]; qv += (double)r[1];
      }
      const long long vs = (long long)a.Do * a.Ho * a.Wo, mt = lid / nbn, tps = vs / M_T;
      const long long b = mt / tps, sp = mt % tps + a.stats_sp0;
      double* o = a.stats + ((b * a.stats_S + sp) * a.Cout + n0 + tid) * 2;
      o[0] = sv; o[1] = qv;
    }
  }
}

// ----------------------------------------------------------------------------------------------
// per-(b,channel) sum / sum-of-squares partials in f64, deterministic two-level reduction
// x: (B,V,C) ; partial: (B,S,C,2) f64 ; WG (b, split) covers rows [split*rows_per, ...)
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                         int V, int C, int S) {
  __shared__ double red[256 * 8];
  const int b = blockIdx.y, sp = blockIdx.x, tid = threadIdx.x;
  const int cg = C / 4;        // float4 groups per row (C <= 1024 -> cg <= 256)
  const int rpp = 256 / cg;    // rows per pass
  const int rows_per = (V + S - 1) / S;
  const int v0 = sp * rows_per, v1 = min(V, v0 + rows_per);
  const int g = tid % cg, ro = tid / cg;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (ro < rpp)
    for (int v = v0 + ro; v < v1; v += rpp) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(x + ((long long)b * V + v) * C + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] += (double)t[j]; q[j] += (double)t[j] * (double)t[j]; }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[tid * 8 + j] = s[j]; red[tid * 8 + 4 + j] = q[j]; }
  __syncthreads();
  if (ro == 0) {
    for (int r = 1; r < rpp; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j] += red[(tid + r * cg) * 8 + j]; q[j] += red[(tid + r * cg) * 8 + 4 + j]; }
    double* o = partial + (((long long)b * S + sp) * C + 4 * g) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = s[j]; o[2 * j + 1] = q[j]; }
  }
}

// GroupNorm(groups, eps) coefficients: scale[b,c] = gamma[c]*rstd[b,g], shift[b,c] = beta[c]-mean*scale.
// One workgroup per shape, one WAVEFRONT per group (groups beyond the 4 waves are taken in turns): the cpg x S f64 partials of a group
// are summed lane-parallel in a fixed pattern (lane l takes items l, l + 64, ...; then a shuffle tree): deterministic, and 64 x
// shorter than the one-thread-per-group loop it replaces (46 us per call, 20 calls per decoded batch).
__global__ __launch_bounds__(256) void gn_coeffs_kernel(const double* __restrict__ partial, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                                        int V, int C, int S, int groups, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpg = C / groups;
  for (int g = wave; g < groups; g += 4) {
    double s = 0, q = 0;
    for (int it = lane; it < cpg * S; it += 64) {
      const int c = g * cpg + it / S, sp = it % S;
      const double* p = partial + (((long long)b * S + sp) * C + c) * 2;
      s += p[0]; q += p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    const double n = (double)V * cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
      const float sc = gamma[c] * rstd;
      scale[b * C + c] = sc;
      shift[b * C + c] = beta[c] - (float)mean * sc;
    }
  }
}

__global__ void affine_cl_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                 const float* __restrict__ shift, float* __restrict__ y, long long V, int C, long long total4) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int cg = C / 4;
  const int c4 = (int)(i % cg);
  const long long row = i / cg;
  const int b = (int)(row / V);
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 s = *reinterpret_cast<const f32x4*>(scale + (long long)b * C + 4 * c4);
  const f32x4 t = *reinterpret_cast<const f32x4*>(shift + (long long)b * C + 4 * c4);
  reinterpret_cast<f32x4*>(y)[i] = v * s + t;
}

__global__ void maxpool2_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int Do, int Ho, int Wo, int C) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cg = C / 4;
  const long long total = (long long)B * Do * Ho * Wo * cg;
  if (i >= total) return;
  const int c4 = (int)(i % cg); long long r = i / cg;
  const int xo = (int)(r % Wo); r /= Wo;
  const int yo = (int)(r % Ho); r /= Ho;
  const int zo = (int)(r % Do); const int b = (int)(r / Do);
  const int Di = 2 * Do, Hi = 2 * Ho, Wi = 2 * Wo;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const long long off = ((((long long)b * Di + 2 * zo + (d >> 2)) * Hi + 2 * yo + ((d >> 1) & 1)) * Wi + 2 * xo + (d & 1)) * C + 4 * c4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
    m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
  }
  reinterpret_cast<f32x4*>(y)[i] = m;
}

// out (B,D,H,W,Cs+Cu) = cat[ skip (B,D,H,W,Cs), nearest_x2( low (B,D/2,H/2,W/2,Cu) ) ]  (unet3d.py:268-283)
__global__ void upcat_cl_kernel(const float* __restrict__ skip, const float* __restrict__ low, float* __restrict__ y,
                                int B, int D, int H, int W, int Cs, int Cu) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int C = Cs + Cu, cg = C / 4;
  const long long total = (long long)B * D * H * W * cg;
  if (i >= total) return;
  const int c = 4 * (int)(i % cg); long long r = i / cg;
  const int xo = (int)(r % W); r /= W;
  const int yo = (int)(r % H); r /= H;
  const int zo = (int)(r % D); const int b = (int)(r / D);
  f32x4 v;
  if (c < Cs)
    v = *reinterpret_cast<const f32x4*>(skip + ((((long long)b * D + zo) * H + yo) * W + xo) * Cs + c);
  else
    v = *reinterpret_cast<const f32x4*>(low + ((((long long)b * (D / 2) + (zo >> 1)) * (H / 2) + (yo >> 1)) * (W / 2) + (xo >> 1)) * Cu + (c - Cs));
  reinterpret_cast<f32x4*>(y)[i] = v;
}

// tiles of M_T output voxels per shape the statistics-capable instances cut a launch into (0: this geometry has no such instance and
// the caller takes the separate statistics pass): the x-reuse forms of the Upsampler's 64- and 32-channel layers
static int conv_stats_tile(const ConvArgs& a) {
  const long long vs = (long long)a.Do * a.Ho * a.Wo, M = vs * a.B;
  const bool xr = g_sfmi_tune.conv_xreuse && a.stride == 1 && !a.up && (a.KS == 2 || a.KS == 3) && a.Wo == a.Wi && a.Ho == a.Hi && a.Do == a.Di &&
                  (a.sp_on || 2 * a.pad == a.KS - 1) && 256 % a.Wo == 0;
  if (!xr || a.resid || a.out_group) return 0;
  const int arows256 = (256 / a.Wo) * (a.Wo + a.KS - 1);
  if (a.Cout == 64 && vs % 256 == 0 && (size_t)(2 * arows256 + 2 * 3 * 64) * LDS_STRIDE * 4 <= 96 * 1024) return 256;
  if (a.Cout == 32 && g_sfmi_tune.conv_xreuse >= 2 && 512 % a.Wo == 0 && vs % 512 == 0 && (M / 512) >= 1024 &&
      (size_t)(2 * (((512 / a.Wo) * (a.Wo + a.KS - 1) + 15) & ~15) + 2 * 3 * 32) * 16 * 4 <= 80 * 1024) return 512;
  return 0;
}

static int conv_dispatch(const ConvArgs& a, void* stream) {
  const int Cout = a.Cout;
  const long long M = (long long)a.B * a.Do * a.Ho * a.Wo;
  hipStream_t st = (hipStream_t)stream;
  // x reuse (see the kernel): stride-1 k2 / k3 convolutions with same-size output whose voxel tile is whole x-rows.  conv_xreuse 1:
  // the 32- and 64-channel output tiles (the Upsampler's four layers); 2 (default): also UNet3D's layers (output channels a multiple
  // of 128) on the 128 x 64 tile with blocked accumulation (ACC2 in the kernel); 3: those on the 128 x 128 swizzled-row tile instead
  // where there are >= 512 of them (5 % faster, the un-blocked MFMA chain's rounding noise).
  const bool xr_geom = g_sfmi_tune.conv_xreuse && a.stride == 1 && !a.up && (a.KS == 2 || a.KS == 3) && a.Wo == a.Wi && a.Ho == a.Hi &&
                       a.Do == a.Di && (a.sp_on || 2 * a.pad == a.KS - 1);
  const bool xr = xr_geom && 256 % a.Wo == 0 && Cout % 128 != 0;
  auto lds_bytes = [&](int M_T, int N_T, bool x) {
    const int arows = x ? (M_T / a.Wo) * (a.Wo + a.KS - 1) : M_T;
    if (x && (N_T == 128 || M_T == 512)) return (size_t)(2 * ((arows + 15) & ~15) + 2 * 3 * N_T) * 16 * 4;   // the swizzled-row instances
    return (size_t)(2 * arows + 2 * (x ? 3 : 1) * N_T) * LDS_STRIDE * 4;
  };
  if (a.stats) {      // only through the *_stats entry points, which checked conv_stats_tile first
    const int mt = conv_stats_tile(a);
    if (mt == 256) {
      static const hipError_t at = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<2, 1, 4, false, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      if (at != hipSuccess) return SFMI_ELDS;
      hipLaunchKernelGGL((conv3d_igemm_kernel<2, 1, 4, false, true, 2, true>), dim3((unsigned)(M / 256)), dim3(256), lds_bytes(256, 64, true), st, a);
    } else if (mt == 512) {
      static const hipError_t at = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<1, 1, 4, false, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      if (at != hipSuccess) return SFMI_ELDS;
      hipLaunchKernelGGL((conv3d_igemm_kernel<1, 1, 4, false, true, 4, true>), dim3((unsigned)(M / 512)), dim3(256), lds_bytes(512, 32, true), st, a);
    } else return SFMI_EINVAL;
    SFMI_CHECK_LAUNCH();
    return SFMI_OK;
  }
  if (Cout % 128 == 0) {
    constexpr int M_T = 128, N_T = 128;
    const long long tiles = ((M + M_T - 1) / M_T) * (Cout / N_T);
    const bool xr128 = xr_geom && g_sfmi_tune.conv_xreuse >= 2 && 128 % a.Wo == 0;
    if ((tiles < 512 || (xr128 && g_sfmi_tune.conv_xreuse == 2)) && !a.up && g_sfmi_tune.conv_xreuse >= 2) {
      // 128 x 64 tiles: the blocked-accumulation form; also what a coarse grid (4^3, 8^3 x 128 channels: fewer 128 x 128 tiles than
      // the 512 workgroups the chip holds) takes in every mode
      constexpr int N_H = 64;
      dim3 grid((unsigned)(((M + M_T - 1) / M_T) * (Cout / N_H)));
      static const hipError_t attrh = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<1, 2, 2, false, true>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      if (xr128 && attrh == hipSuccess && lds_bytes(M_T, N_H, true) <= 80 * 1024)
        hipLaunchKernelGGL((conv3d_igemm_kernel<1, 2, 2, false, true>), grid, dim3(256), lds_bytes(M_T, N_H, true), st, a);
      else hipLaunchKernelGGL((conv3d_igemm_kernel<1, 2, 2, false>), grid, dim3(256), lds_bytes(M_T, N_H, false), st, a);
      SFMI_CHECK_LAUNCH();
      return SFMI_OK;
    }
    dim3 grid((unsigned)tiles);
    static const hipError_t attr128 = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<2, 2, 2, false, true>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (xr128 && attr128 == hipSuccess && lds_bytes(M_T, N_T, true) <= 80 * 1024)
      hipLaunchKernelGGL((conv3d_igemm_kernel<2, 2, 2, false, true>), grid, dim3(256), lds_bytes(M_T, N_T, true), st, a);
    else if (a.up) hipLaunchKernelGGL((conv3d_igemm_kernel<2, 2, 2, true>), grid, dim3(256), lds_bytes(M_T, N_T, false), st, a);
    else hipLaunchKernelGGL((conv3d_igemm_kernel<2, 2, 2, false>), grid, dim3(256), lds_bytes(M_T, N_T, false), st, a);
  } else if (Cout % 64 == 0) {
    constexpr int M_T = 256, N_T = 64;
    dim3 grid((unsigned)(((M + M_T - 1) / M_T) * (Cout / N_T)));
    // the framed x-rows of a narrow grid (Wo <= 2: 112 KB) do not fit the 96 KB the x-reuse instance may ask for, and the device
    // may refuse the raised limit: both fall through to the per-tap form instead of failing the call
    static const hipError_t attr64 = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<2, 1, 4, false, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (xr && attr64 == hipSuccess && lds_bytes(M_T, N_T, true) <= 96 * 1024)
      hipLaunchKernelGGL((conv3d_igemm_kernel<2, 1, 4, false, true>), grid, dim3(256), lds_bytes(M_T, N_T, true), st, a);
    else if (a.up) hipLaunchKernelGGL((conv3d_igemm_kernel<2, 1, 4, true>), grid, dim3(256), lds_bytes(M_T, N_T, false), st, a);
    else hipLaunchKernelGGL((conv3d_igemm_kernel<2, 1, 4, false>), grid, dim3(256), lds_bytes(M_T, N_T, false), st, a);
  } else {
    constexpr int M_T = 256, N_T = 32;
    dim3 grid((unsigned)(((M + M_T - 1) / M_T) * (Cout / N_T)));
    // 512-voxel tiles (four 32-voxel tiles per wave) where the grid is wide enough for their framed rows to fit twice per CU
    // (Wo >= 32) and there are enough of them to fill the chip; conv_xreuse 1 keeps the round-4 256-voxel tile
    static const hipError_t attr32w = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<1, 1, 4, false, true, 4>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (xr && g_sfmi_tune.conv_xreuse >= 2 && attr32w == hipSuccess && 512 % a.Wo == 0 && lds_bytes(512, N_T, true) <= 80 * 1024 &&
        ((long long)a.Do * a.Ho * a.Wo) % 512 == 0 && (M / 512) * (Cout / N_T) >= 1024) {
      dim3 gridw((unsigned)(((M + 511) / 512) * (Cout / N_T)));
      hipLaunchKernelGGL((conv3d_igemm_kernel<1, 1, 4, false, true, 4>), gridw, dim3(256), lds_bytes(512, N_T, true), st, a);
      SFMI_CHECK_LAUNCH();
      return SFMI_OK;
    }
    static const hipError_t attr32 = hipFuncSetAttribute((const void*)conv3d_igemm_kernel<1, 1, 4, false, true>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (xr && attr32 == hipSuccess && lds_bytes(M_T, N_T, true) <= 96 * 1024)      // Wo == 1 needs 138 KB: per-tap form
      hipLaunchKernelGGL((conv3d_igemm_kernel<1, 1, 4, false, true>), grid, dim3(256), lds_bytes(M_T, N_T, true), st, a);
    else if (a.up) hipLaunchKernelGGL((conv3d_igemm_kernel<1, 1, 4, true>), grid, dim3(256), lds_bytes(M_T, N_T, false), st, a);
    else hipLaunchKernelGGL((conv3d_igemm_kernel<1, 1, 4, false>), grid, dim3(256), lds_bytes(M_T, N_T, false), st, a);
  }
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

extern "C" {

// host: torch Conv3d weight (Cout,Cin,k,k,k) -> [tap][Cout][Cin]
int sfmi_conv_pack_weight(const float* w, int Cout, int Cin, int KS, float* out) {
  if (!w || !out) return SFMI_EINVAL;
  const int T = KS * KS * KS;
  for (int co = 0; co < Cout; ++co)
    for (int ci = 0; ci < Cin; ++ci)
      for (int t = 0; t < T; ++t) out[((size_t)t * Cout + co) * Cin + ci] = w[((size_t)co * Cin + ci) * T + t];
  return SFMI_OK;
}

// replaces nn.Conv3d (+ fused input GroupNorm-apply, nearest-x2 upsample, bias, ReLU); see file header.
// x (B,Di,Hi,Wi,Cin) -> y (B,Do,Ho,Wo,Cout), Do = ((Di<<up) + 2*pad - KS)/stride + 1.
int sfmi_conv3d_cl_stats_f32(const float* x, const float* wT, const float* in_scale, const float* in_shift, const float* bias, float* y, int B,
                             int Di, int Hi, int Wi, int Cin, int Cout, int KS, int stride, int pad, int up, int relu, double* partial, int* splits,
                             void* stream);
int sfmi_conv3d_cl_f32(const float* x, const float* wT, const float* in_scale, const float* in_shift,
                       const float* bias, float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int KS,
                       int stride, int pad, int up, int relu, void* stream) {
  return sfmi_conv3d_cl_stats_f32(x, wT, in_scale, in_shift, bias, y, B, Di, Hi, Wi, Cin, Cout, KS, stride, pad, up, relu, nullptr, nullptr, stream);
}
// the same; partial != NULL: also the GroupNorm statistics of y as per-(shape, split, channel) f64 partials (see sfmi_conv3d_up2_cl_stats_f32)
int sfmi_conv3d_cl_stats_f32(const float* x, const float* wT, const float* in_scale, const float* in_shift, const float* bias, float* y, int B,
                             int Di, int Hi, int Wi, int Cin, int Cout, int KS, int stride, int pad, int up, int relu, double* partial, int* splits,
                             void* stream) {
  if ((partial == nullptr) != (splits == nullptr)) return SFMI_EINVAL;
  if (!x || !wT || !y || B <= 0 || Cin % KC || Cout % 32 || (KS != 1 && KS != 2 && KS != 3)) return SFMI_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return SFMI_EINVAL;
  ConvArgs a;
  a.x = x; a.wT = wT; a.in_scale = in_scale; a.in_shift = in_shift; a.bias = bias; a.y = y;
  a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KS = KS; a.stride = stride; a.pad = pad;
  a.up = up; a.relu = relu; a.resid = nullptr; a.out_group = 0; a.out_group_stride = 0;
  a.sp_on = 0; a.spz = a.spy = a.spx = 0; a.padz = a.pady = a.padx = 0; a.stats = nullptr; a.stats_S = 0; a.stats_sp0 = 0;
  a.Do = ((Di << up) + 2 * pad - KS) / stride + 1;
  a.Ho = ((Hi << up) + 2 * pad - KS) / stride + 1;
  a.Wo = ((Wi << up) + 2 * pad - KS) / stride + 1;
  if (partial) {
    const int mt = conv_stats_tile(a);
    if (!mt) return SFMI_EINVAL;          // nothing launched: the caller takes the separate statistics pass
    a.stats = partial; a.stats_S = (int)((long long)a.Do * a.Ho * a.Wo / mt); a.stats_sp0 = 0;
    *splits = a.stats_S;
  }
  return conv_dispatch(a, stream);
}

// Plain row-major GEMM through the same kernel (a 1x1x1 "conv" over M rows):
//   y[remap(m)][n] = act( sum_k x[m][k] * W[n][k] + bias[n] ) + resid[remap(m)][n]
// used by the transformer PREFILL (mingpt.py:46-111 Linear layers at M = B*L_c rows).
int sfmi_gemm_f32(const float* x, const float* W, const float* bias, const float* resid, float* y, long long M, int N,
                  int K, int act, long long out_group, long long out_group_stride, void* stream) {
  if (!x || !W || !y || M <= 0 || K % KC || N % 32 || M > 0x7fffffffLL) return SFMI_EINVAL;
  ConvArgs a;
  a.x = x; a.wT = W; a.in_scale = nullptr; a.in_shift = nullptr; a.bias = bias; a.y = y; a.resid = resid;
  a.out_group = out_group; a.out_group_stride = out_group_stride;
  a.B = 1; a.Di = 1; a.Hi = 1; a.Wi = (int)M; a.Do = 1; a.Ho = 1; a.Wo = (int)M; a.Cin = K; a.Cout = N; a.KS = 1;
  a.stride = 1; a.pad = 0; a.up = 0; a.relu = act;
  a.sp_on = 0; a.spz = a.spy = a.spx = 0; a.padz = a.pady = a.padx = 0; a.stats = nullptr; a.stats_S = 0; a.stats_sp0 = 0;
  return conv_dispatch(a, stream);
}

// Conv3d(k=3, pad=1) over a nearest-x2-upsampled input WITHOUT the 19 redundant taps: the three taps of an axis read only
// two distinct low-resolution voxels, so each of the 8 output parities is a 2^3 convolution of the low-resolution grid with
// pre-summed weights (even parity: [w0, w1+w2] at offsets (-1, 0); odd: [w0+w1, w2] at (0, +1)); zero padding and the
// per-channel input affine commute with the merge.  8/27 of the FLOPs of the direct form; results differ from it by fp32
// rounding of the weight sums only.
// [host] w (Cout,Cin,3,3,3) -> out [parity = 4 pz + 2 py + px][tap = 4 dz + 2 dy + dx][Cout][Cin]
int sfmi_conv_pack_weight_subpixel(const float* w, int Cout, int Cin, float* out) {
  if (!w || !out) return SFMI_EINVAL;
  // 1-D merge: M[p][d][t] = 1 if original tap t contributes to merged tap d of parity p
  static const int Mg[2][2][3] = {{{1, 0, 0}, {0, 1, 1}}, {{1, 1, 0}, {0, 0, 1}}};
  for (int par = 0; par < 8; ++par) {
    const int pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
    for (int tap = 0; tap < 8; ++tap) {
      const int dz = tap >> 2, dy = (tap >> 1) & 1, dx = tap & 1;
      float* o = out + ((size_t)par * 8 + tap) * Cout * Cin;
      for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
          float s = 0.f;
          for (int tz = 0; tz < 3; ++tz)
            for (int ty = 0; ty < 3; ++ty)
              for (int tx = 0; tx < 3; ++tx)
                if (Mg[pz][dz][tz] && Mg[py][dy][ty] && Mg[px][dx][tx]) s += w[(((size_t)co * Cin + ci) * 3 + tz) * 9 + ty * 3 + tx];
          o[(size_t)co * Cin + ci] = s;
        }
    }
  }
  return SFMI_OK;
}

// x (B,Di,Hi,Wi,Cin) low resolution -> y (B,2Di,2Hi,2Wi,Cout) = act(conv3(nearest_x2(affine(x))) + bias); 8 launches
int sfmi_conv3d_up2_cl_stats_f32(const float* x, const float* wsub, const float* in_scale, const float* in_shift, const float* bias,
                                 float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int relu, double* partial, int* splits, void* stream);
int sfmi_conv3d_up2_cl_f32(const float* x, const float* wsub, const float* in_scale, const float* in_shift, const float* bias,
                           float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int relu, void* stream) {
  return sfmi_conv3d_up2_cl_stats_f32(x, wsub, in_scale, in_shift, bias, y, B, Di, Hi, Wi, Cin, Cout, relu, nullptr, nullptr, stream);
}
// the same; partial != NULL: the launches also leave the GroupNorm statistics of y - per (shape, split, channel) f64 {sum, sum of squares},
// *splits of them per shape - for sfmi_groupnorm_coeffs_partial_f32, instead of a statistics pass over y.  SFMI_EINVAL BEFORE anything is
// launched when this geometry has no statistics-capable instance (the caller then takes sfmi_conv3d_up2_cl_f32 + sfmi_groupnorm_coeffs_f32).
int sfmi_conv3d_up2_cl_stats_f32(const float* x, const float* wsub, const float* in_scale, const float* in_shift, const float* bias,
                                 float* y, int B, int Di, int Hi, int Wi, int Cin, int Cout, int relu, double* partial, int* splits, void* stream) {
  if (!x || !wsub || !y || B <= 0 || Cin % KC || Cout % 32 || ((partial == nullptr) != (splits == nullptr))) return SFMI_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return SFMI_EINVAL;
  int tps = 0;
  for (int par = partial ? -1 : 0; par < 8; ++par) {      // par = -1: dry pass that only asks whether the statistics instance exists
    ConvArgs a;
    a.stats = partial; a.stats_S = 8 * tps; a.stats_sp0 = (par < 0 ? 0 : par) * tps;
    a.x = x; a.wT = wsub + (size_t)(par < 0 ? 0 : par) * 8 * Cout * Cin; a.in_scale = in_scale; a.in_shift = in_shift; a.bias = bias; a.y = y;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Cout = Cout; a.KS = 2; a.stride = 1; a.pad = 0; a.up = 0; a.relu = relu;
    a.resid = nullptr; a.out_group = 0; a.out_group_stride = 0;
    a.Do = Di; a.Ho = Hi; a.Wo = Wi;
    const int pp = par < 0 ? 0 : par;
    a.sp_on = 1; a.spz = pp >> 2; a.spy = (pp >> 1) & 1; a.spx = pp & 1;
    a.padz = 1 - a.spz; a.pady = 1 - a.spy; a.padx = 1 - a.spx;
    if (par < 0) {
      const int mt = conv_stats_tile(a);
      if (!mt) return SFMI_EINVAL;
      tps = (int)((long long)Di * Hi * Wi / mt);
      *splits = 8 * tps;
      continue;
    }
    const int rc = conv_dispatch(a, stream);
    if (rc != SFMI_OK) return rc;
  }
  return SFMI_OK;
}

int sfmi_gn_splits(int V) { return V >= 32768 ? 64 : (V >= 4096 ? 16 : (V >= 512 ? 4 : 1)); }

// replaces nn.GroupNorm statistics (unet3d.py:66, updown.py): x (B,V,C) -> scale/shift (B,C) such that
// GN(x) == x*scale + shift.  partial: workspace of B*sfmi_gn_splits(V)*C*2 doubles.
int sfmi_groupnorm_coeffs_f32(const float* x, const float* gamma, const float* beta, float* scale, float* shift,
                              double* partial, int B, int V, int C, int groups, float eps, void* stream) {
  if (!x || !gamma || !beta || !scale || !shift || !partial || C % 4 || C > 1024 || C % groups || groups > 64) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int S = sfmi_gn_splits(V);
  hipLaunchKernelGGL(chan_stats_kernel, dim3(S, B), dim3(256), 0, st, x, partial, V, C, S);
  hipLaunchKernelGGL(gn_coeffs_kernel, dim3(B), dim3(256), 0, st, partial, gamma, beta, scale, shift, V, C, S, groups, eps);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

// GroupNorm coefficients from statistics partials a convolution left behind (sfmi_conv3d_*_stats_f32): partial (B, S, C, 2) f64, V voxels per shape
int sfmi_groupnorm_coeffs_partial_f32(const double* partial, const float* gamma, const float* beta, float* scale, float* shift, int B, int V, int C,
                                      int S, int groups, float eps, void* stream) {
  if (!partial || !gamma || !beta || !scale || !shift || B <= 0 || V <= 0 || S <= 0 || C % 4 || C > 1024 || C % groups || groups > 64) return SFMI_EINVAL;
  hipLaunchKernelGGL(gn_coeffs_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, partial, gamma, beta, scale, shift, V, C, S, groups, eps);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_affine_cl_f32(const float* x, const float* scale, const float* shift, float* y, int B, long long V, int C, void* stream) {
  if (!x || !scale || !shift || !y || C % 4) return SFMI_EINVAL;
  const long long total4 = (long long)B * V * (C / 4);
  hipLaunchKernelGGL(affine_cl_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, V, C, total4);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_maxpool2_cl_f32(const float* x, float* y, int B, int Do, int Ho, int Wo, int C, void* stream) {
  if (!x || !y || C % 4) return SFMI_EINVAL;
  const long long total = (long long)B * Do * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool2_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, B, Do, Ho, Wo, C);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

int sfmi_upcat_cl_f32(const float* skip, const float* low, float* y, int B, int D, int H, int W, int Cs, int Cu, void* stream) {
  if (!skip || !low || !y || Cs % 4 || Cu % 4) return SFMI_EINVAL;
  const long long total = (long long)B * D * H * W * ((Cs + Cu) / 4);
  hipLaunchKernelGGL(upcat_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, skip, low, y, B, D, H, W, Cs, Cu);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
