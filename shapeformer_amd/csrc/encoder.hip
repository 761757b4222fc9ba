// VQDIF LocalPoolPointnet per-point path for gfx950 (reference: shapeformer/models/vqdif/enc.py:95-140,
// layers.py:39-48, common.py:260-321; torch_scatter.scatter_max / scatter_mean call sites enc.py:70-74,103-110).
//
// MI355X design (not the reference's dense-grid dataflow):
//   * the 64^3 cell id is computed ONCE (it is identical for the 4 pooling passes and the mean);
//   * the points of a shape are GROUPED BY CELL once (counting sort: per-cell histogram on a 1 MB/shape int map, exclusive
//     scan, scatter - two integer atomics per point in total); every later pass walks the points in that order, so the points
//     of a cell are consecutive lanes of a wave: the local max pool and the mean are SEGMENTED WAVEFRONT REDUCTIONS (5 shuffle
//     steps over the 32 points of a tile) and only the last lane of each run touches memory (one atomic per run and channel
//     instead of one per point and channel: runs that cross a tile boundary are the only contended ones);
//   * a cell's "segment" is named by the sorted position of its first point, so the per-pass pooled maxima live in a
//     compact (B,T,32) buffer instead of a 33.5 MB/shape dense grid;
//   * max is order-independent -> integer-ordered max gives bit-exact, deterministic results whatever the order inside a cell;
//   * the grid mean accumulates in 2^-32 fixed point (int64): associative, hence deterministic and order-independent, and
//     more accurate than an f32 running sum;
//   * the per-point MLP blocks run on f32 MFMA (32x32x2) with activations in the C/D register layout
//     (same chaining trick as sdf_query.hip), weights staged per workgroup in LDS.
#include "sfmi_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define ENC_G 64

// packed per-block weights (floats): [fc0_lo 1024][fc0_hi 1024][sc_lo 1024][sc_hi 1024][fc1 1024][b0 32][b1 32]
#define ENC_BLK_FLOATS (5 * 1024 + 64)
// block-0 kernel additionally needs fc_pos: [half][2][64] = 256 floats ; last block needs fc_c: 1024 + 32
#define ENC_OFF_BLK(k) ((k) * ENC_BLK_FLOATS)
#define ENC_OFF_FCPOS (5 * ENC_BLK_FLOATS)
#define ENC_OFF_FCC (ENC_OFF_FCPOS + 256)
#define ENC_PACK_FLOATS (ENC_OFF_FCC + 1024 + 32)

__device__ __forceinline__ float relu(float x) { return fmaxf(x, 0.0f); }

// monotone float <-> int key (signed compare order == float order, NaN-free inputs)
__device__ __forceinline__ int fkey(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// ---------------------------------------------------------------------------------------------
// E0: cell ids (a2,a3), per-cell point counts, latent occupancy mask (enc.py:85-91)
// ---------------------------------------------------------------------------------------------
__global__ void enc_cells_kernel(const float* __restrict__ cloud,  // (B,T,3) in [-1,1]
                                 int* __restrict__ cell,            // (B,T)
                                 int* __restrict__ cnt,             // (B,G^3) pre-zeroed: points per cell
                                 unsigned char* __restrict__ mask,  // (B,R,R,R) pre-zeroed, [z][y][x]
                                 int B, int T, int R) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * T) return;
  int b = (int)(i / T), t = (int)(i - (long long)b * T);
  const float* p = cloud + i * 3;
  float ux = sfmi_normalize(p[0] * 0.5f), uy = sfmi_normalize(p[1] * 0.5f), uz = sfmi_normalize(p[2] * 0.5f);
  int cx = (int)(ux * (float)ENC_G), cy = (int)(uy * (float)ENC_G), cz = (int)(uz * (float)ENC_G);
  int c = cx + ENC_G * (cy + ENC_G * cz);  // common.py:300-321 'original' order
  cell[i] = c;
  atomicAdd(&cnt[(long long)b * ENC_G * ENC_G * ENC_G + c], 1);
  int mx = (int)(ux * (float)R), my = (int)(uy * (float)R), mz = (int)(uz * (float)R);
  mask[(((long long)b * R + mz) * R + my) * R + mx] = 1;
}

// E0b: exclusive scan of one shape's 64^3 cell counts -> start[cell] (sorted position of the cell's first point), and a copy
// `cursor` that the scatter advances.  Two coalesced launches over 4096-cell chunks (64 per shape): chunk sums, then every chunk
// scans itself on top of the sum of the chunks before it (a 64-entry prefix formed by one wave) - 64 x B workgroups instead of B
// (round 3's one-workgroup-per-shape scan with 256 consecutive cells per THREAD was uncoalesced and took 150-250 us).
constexpr int ENC_CH = 4096, ENC_NCH = ENC_G * ENC_G * ENC_G / ENC_CH;     // 64 chunks per shape
__global__ __launch_bounds__(1024) void enc_scan_sums_kernel(const int* __restrict__ cnt, int* __restrict__ chunk_sum) {
  __shared__ int wsum[16];
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int4 v = *reinterpret_cast<const int4*>(cnt + ((long long)b * ENC_NCH + ch) * ENC_CH + 4 * tid);
  int t = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
  if (lane == 0) wsum[wave] = t;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += wsum[w];
    chunk_sum[b * ENC_NCH + ch] = tot;
  }
}
__global__ __launch_bounds__(1024) void enc_scan_apply_kernel(int* __restrict__ start /*in: counts*/, int* __restrict__ cursor,
                                                              const int* __restrict__ chunk_sum, int* __restrict__ flag, int limit) {
  __shared__ int wsum[16];
  __shared__ int s_base;
  const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long off = ((long long)b * ENC_NCH + ch) * ENC_CH + 4 * tid;
  const int4 v = *reinterpret_cast<const int4*>(start + off);
  if (wave == 0) {          // cells of the chunks before this one
    int c = lane < ch ? chunk_sum[b * ENC_NCH + lane] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) s_base = c;
  }
  // a cell with more points than the fused stage kernel's workgroups can hold: the staged kernels take the call (see enc_fused_kernel)
  if (__any(max(max(v.x, v.y), max(v.z, v.w)) > limit) && lane == 0) { atomicOr(flag, 1); atomicOr(flag + 1 + b, 1); }
  const int tot = (v.x + v.y) + (v.z + v.w);
  int incl = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = s_base + incl - tot;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int4 o;
  o.x = base; o.y = o.x + v.x; o.z = o.y + v.y; o.w = o.z + v.z;
  *reinterpret_cast<int4*>(start + off) = o;
  *reinterpret_cast<int4*>(cursor + off) = o;
}

// E0c: scatter the points into cell order: order[b][pos] = t, scell[b][pos] = cell (any order INSIDE a cell: max and the
// fixed-point mean do not depend on it)
__global__ void enc_scatter_kernel(const int* __restrict__ cell, int* __restrict__ cursor, int* __restrict__ order,
                                   int* __restrict__ scell, int B, int T) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * T) return;
  const int b = (int)(i / T), t = (int)(i - (long long)b * T), c = cell[i];
  const int pos = atomicAdd(&cursor[(long long)b * ENC_G * ENC_G * ENC_G + c], 1);
  order[(long long)b * T + pos] = t;
  scell[(long long)b * T + pos] = c;
}

// debug taps only: rows of a (B,T,C) buffer from cell order back to the caller's point order
__global__ void enc_unsort_kernel(const float* __restrict__ src, const int* __restrict__ order, float* __restrict__ dst, int B, int T, int C) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = gid / C;
  if (i >= (long long)B * T) return;
  const int b = (int)(i / T);
  dst[((long long)b * T + order[i]) * C + gid % C] = src[gid];
}

// one 32x32 layer: acc += W * x  (x in C/D layout, optional ReLU on the fly)
template <bool RELU>
__device__ __forceinline__ void layer32(const f32x4* __restrict__ Wl, const f32x16& x, f32x16& acc) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 w = Wl[64 * g];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = MFMA(w[j], RELU ? relu(x[4 * g + j]) : x[4 * g + j], acc);
  }
}

__device__ __forceinline__ void bias_init(const float* __restrict__ b, int hi, f32x16& acc) {
  const f32x4* bp = reinterpret_cast<const f32x4*>(b + 4 * hi);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = bp[2 * g];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
  }
}

// ---------------------------------------------------------------------------------------------
// E1..E5: one ResnetBlockFC stage per launch (the pooling between stages is a global dependency).
//   STAGE 0 : x = fc_pos(p)                      (enc.py:125-127)
//   STAGE k : x = cat[net_{k-1}, pooled_{k-1}]   (enc.py:128-131)
//   out net_k ; segmented wavefront max -> segmax_k (k<4) ; k==4: c = fc_c(net) -> segmented fixed-point sums for the mean
// Points are processed in CELL ORDER (order / scell): a tile = 32 consecutive sorted points, lane pl = point.
// ---------------------------------------------------------------------------------------------
template <int STAGE>
__global__ __launch_bounds__(256) void enc_block_kernel(
    const float* __restrict__ cloud, const int* __restrict__ cell /*sorted*/, const int* __restrict__ rep /*start[cell]*/,
    const int* __restrict__ order,        // (B,T) sorted position -> point index
    const float* __restrict__ net_in,     // (B,T,32) in sorted order
    const int* __restrict__ segmax_in,    // (B,T,32) ordered-int keys
    float* __restrict__ net_out,          // (B,T,32)
    int* __restrict__ segmax_out,         // (B,T,32) pre-filled 0x80808080
    long long* __restrict__ csum,         // (B,T,32) pre-zeroed   (STAGE 4)
    int* __restrict__ ccount,             // (B,T)    pre-zeroed   (STAGE 4)
    const float* __restrict__ wpack, int B, int T,
    const int* __restrict__ run_flag) {   // optional: [0] any shape, [1 + b] shape b was declined by the fused kernel - only those run
  if (run_flag && !run_flag[0]) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // stage weights: block STAGE (+ fc_pos for 0, + fc_c for 4)
  {
    const f32x4* s = reinterpret_cast<const f32x4*>(wpack + ENC_OFF_BLK(STAGE));
    f32x4* d = reinterpret_cast<f32x4*>(lds);
    for (int i = threadIdx.x; i < ENC_BLK_FLOATS / 4; i += blockDim.x) d[i] = s[i];
    if (STAGE == 0) {
      const f32x4* s2 = reinterpret_cast<const f32x4*>(wpack + ENC_OFF_FCPOS);
      f32x4* d2 = reinterpret_cast<f32x4*>(lds + ENC_BLK_FLOATS);
      for (int i = threadIdx.x; i < 256 / 4; i += blockDim.x) d2[i] = s2[i];
    }
    if (STAGE == 4) {
      const f32x4* s2 = reinterpret_cast<const f32x4*>(wpack + ENC_OFF_FCC);
      f32x4* d2 = reinterpret_cast<f32x4*>(lds + ENC_BLK_FLOATS);
      for (int i = threadIdx.x; i < (1024 + 32) / 4; i += blockDim.x) d2[i] = s2[i];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, hi = lane >> 5, pl = lane & 31;
  const long long total = (long long)B * T;
  const long long ntiles = (total + 31) >> 5;
  const long long wave_gid = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  const f32x4* L = reinterpret_cast<const f32x4*>(lds) + lane;

  for (long long tile = wave_gid; tile < ntiles; tile += nwaves) {
    long long i = tile * 32 + pl;
    bool valid = i < total;
    if (!valid) i = total - 1;
    const int b = (int)(i / T);
    if (run_flag && !run_flag[1 + b]) valid = false;
    // segment = sorted position of the cell's first point (b*T + start[cell] < 2^31); for the run tests a lane that is not
    // valid gets a segment of its own
    const int segi = (int)((long long)b * T) + rep[(long long)b * ENC_G * ENC_G * ENC_G + cell[i]];
    const long long seg = segi;
    const int segq = valid ? segi : -1 - pl;
    // run structure of the tile (both halves of the wave hold the same 32 points): same run as the lane `off` below?
    // (every shuffle is executed by ALL lanes - a short-circuited `&&` would read from lanes that skipped it)
    bool same[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) { const int o_ = __shfl_up(segq, 1 << k, 32); same[k] = pl >= (1 << k) && o_ == segq; }
    const int nxt_ = __shfl_down(segq, 1, 32);
    const bool run_last = pl == 31 || nxt_ != segq;

    f32x16 xlo, xhi;
    if (STAGE == 0) {
      const float* p = cloud + ((long long)b * T + order[i]) * 3;
      const float hx = p[0] * 0.5f, hy = p[1] * 0.5f, hz = p[2] * 0.5f;  // vqdif.py:36 Xbd/2
      const float* fp = lds + ENC_BLK_FLOATS;
#pragma unroll
      for (int t = 0; t < 16; ++t) { xlo[t] = 0.0f; xhi[t] = 0.0f; }
      xlo = MFMA(fp[lane], hi ? hy : hx, xlo);
      xlo = MFMA(fp[64 + lane], hi ? 1.0f : hz, xlo);
      xhi = MFMA(fp[128 + lane], hi ? hy : hx, xhi);
      xhi = MFMA(fp[192 + lane], hi ? 1.0f : hz, xhi);
    } else {
      const f32x4* np_ = reinterpret_cast<const f32x4*>(net_in + i * 32 + 4 * hi);
      const int4* sp = reinterpret_cast<const int4*>(segmax_in + seg * 32 + 4 * hi);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = np_[2 * g];
        int4 k = sp[2 * g];
        xlo[4 * g + 0] = v[0]; xlo[4 * g + 1] = v[1]; xlo[4 * g + 2] = v[2]; xlo[4 * g + 3] = v[3];
        xhi[4 * g + 0] = fkey_inv(k.x); xhi[4 * g + 1] = fkey_inv(k.y);
        xhi[4 * g + 2] = fkey_inv(k.z); xhi[4 * g + 3] = fkey_inv(k.w);
      }
    }
    // h = fc_0(relu(x)) + b0 ; out = shortcut(x) + fc_1(relu(h)) + b1      (layers.py:39-48)
    f32x16 h, o;
    bias_init(lds + 5 * 1024, hi, h);
    layer32<true>(L + 0 * 256, xlo, h);
    layer32<true>(L + 1 * 256, xhi, h);
    bias_init(lds + 5 * 1024 + 32, hi, o);
    layer32<false>(L + 2 * 256, xlo, o);
    layer32<false>(L + 3 * 256, xhi, o);
    layer32<true>(L + 4 * 256, h, o);

    if (STAGE < 4) {
      if (valid) {
        f32x4* op = reinterpret_cast<f32x4*>(net_out + i * 32 + 4 * hi);
        int* sm = segmax_out + seg * 32 + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
          op[2 * g] = v;
        }
      }
      // local max pool (enc.py:95-112): segmented inclusive max over the points of the tile, then ONE atomic per run and channel
      int key[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) key[t] = fkey(o[t]);
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int u = __shfl_up(key[t], 1 << k, 32); if (same[k]) key[t] = max(key[t], u); }
      if (valid && run_last) {
        int* sm = segmax_out + seg * 32 + 4 * hi;
        if (pl < 31 && seg >= tile * 32) {
          // the run starts AND ends inside this tile: no other wave ever touches its segment, the pooled maxima are final -
          // four 16-byte stores instead of sixteen atomics on one cache line (most runs: a cell holds ~5 points)
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<int4*>(sm + 8 * g) = int4{key[4 * g], key[4 * g + 1], key[4 * g + 2], key[4 * g + 3]};
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicMax(sm + 8 * g + j, key[4 * g + j]);
        }
      }
    } else {
      // c = fc_c(net) (enc.py:133) -> fixed-point accumulate for the per-cell mean (enc.py:70-74)
      f32x16 c;
      bias_init(lds + ENC_BLK_FLOATS + 1024, hi, c);
      layer32<false>(reinterpret_cast<const f32x4*>(lds + ENC_BLK_FLOATS) + lane, o, c);
      if (valid && net_out) {   // debug tap (sfmi_encode_points_tap_f32): row i = [block-4 output (32) | c = fc_c(net) (32)]
        f32x4* tp = reinterpret_cast<f32x4*>(net_out + ((long long)b * T + order[i]) * 64 + 4 * hi);   // caller's point order
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          tp[2 * g] = f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
          tp[8 + 2 * g] = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
        }
      }
      // scatter_mean numerator (enc.py:70-74): segmented inclusive SUM in 2^-32 fixed point, one atomic per run and channel
      long long q[16];
      int n = valid ? 1 : 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) q[t] = valid ? __double2ll_rn((double)c[t] * 4294967296.0) : 0ll;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int lo = __shfl_up((int)(q[t] & 0xffffffffll), 1 << k, 32), hi32 = __shfl_up((int)(q[t] >> 32), 1 << k, 32);
          if (same[k]) q[t] += ((long long)hi32 << 32) | (unsigned int)lo;
        }
        const int un = __shfl_up(n, 1 << k, 32);
        if (same[k]) n += un;
      }
      if (valid && run_last) {
        long long* sp = csum + seg * 32 + 4 * hi;
        if (pl < 31 && seg >= tile * 32) {      // run wholly inside the tile (see the max pool above): plain stores into the zeroed sums
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; j += 2) *reinterpret_cast<longlong2*>(sp + 8 * g + j) = longlong2{q[4 * g + j], q[4 * g + j + 1]};
          if (hi == 0) ccount[seg] = n;
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(reinterpret_cast<unsigned long long*>(sp + 8 * g + j), (unsigned long long)q[4 * g + j]);
          if (hi == 0) atomicAdd(ccount + seg, n);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// E1..E5 in ONE launch (round 5): run-aligned workgroups.
// The only dependency between the five stages is the local max pool, and a pool never leaves its cell.  A workgroup that owns
// WHOLE cells can therefore walk all five stages by itself: the running features of its points stay in registers (one 32-point
// tile per wave, sixteen waves), the pooled maxima of its cells live in 72 KB of LDS (one row per sorted position, named like the
// global segments: the cell's first point), a stage's weights are staged per stage (fetched into registers one stage ahead), and
// three workgroup barriers per stage replace the launch boundary.  Nothing but the cloud is read from HBM and nothing but the
// per-cell sums written: the four (B,T,32) feature round trips, the four pooled-maxima buffers and their 0x80 fills of the staged
// form are gone.
// Ownership: workgroup w of a shape takes the runs that START in sorted positions [w * EF_NOM, (w + 1) * EF_NOM) - it skips the
// tail of a run begun before and finishes the run that straddles its end - which holds at most EF_NOM + EF_LIMIT - 1 <= EF_CAP
// points if no cell has more than EF_LIMIT = 128 of them (the bench's synthetic partial clouds: 34 .. 52).  The scan kernel raises
// flag[1 + shape] (and flag[0]) when one does: this kernel then skips the shape and the staged kernels, launched behind it with the
// opposite test, take it - no host round trip either way.  Per point the arithmetic is the staged form's, instruction for
// instruction (an MFMA column depends on its own point only), the pool is an integer max and the mean a fixed-point sum: the two
// forms are BIT-IDENTICAL.
// Measured (profiles/r05_kbench_enc.txt): 0.9 ms per 64 x 16 384 points = 62 TFLOP/s of f32 MFMA against 0.42 ms x 5 for the staged
// kernels.  PMC (round 5): MFMA busy 0.40, 45 % of the wave cycles waiting on an instruction dependency, 13 % on LDS.  Tried without
// effect on that figure: eight waves x two tiles (254 VGPRs), 256-point workgroups (two per CU), segmented wavefront scans instead
// of the LDS atomics, weights fetched between the barriers, the operand ReLUs hoisted out of the MFMA chains (1.46 -> 1.44 ms for
// the call, the staged kernels 2.51 -> 2.57: not kept).
// ---------------------------------------------------------------------------------------------
constexpr int EF_CAP = 512, EF_LIMIT = 128, EF_NOM = EF_CAP - EF_LIMIT, EF_NT = 1, EF_THREADS = EF_CAP / (32 * EF_NT) * 64;
constexpr int EF_W_FLOATS = ENC_BLK_FLOATS + 1056;      // stage weights (+ fc_pos or fc_c)
// row stride of the pooled maxima (ints) and of the fixed-point sums (64-bit words): 32 channels + 4 of padding.  With 32 the points of
// a tile - a handful of different cells - fell on two banks per channel and every LDS atomic was a 16-way bank conflict.
constexpr int EF_ROW = 36;

struct EfTile {
  int i;          // global sorted position b * T + pos (clamped for invalid lanes)
  int seg_l;      // sorted position of the cell's first point relative to the workgroup's first point: row of the LDS pooled maxima
  bool valid;
  bool any;       // wave-uniform: the tile holds a point at all (a workgroup's last tiles are usually empty: n is 384 .. 511 of 512)
};

// The pools are LDS atomics, one per point and channel (ds_max_i32 on the cell's row; the points of a cell collide and are
// serialised by the LDS - a few cycles): the first version carried the staged kernels' segmented wavefront scans over, 80
// ds_bpermute per tile and stage each waited for on its own, and spent more time in them than in its MFMAs.
// a stage's weights travel global -> registers during the stage before, registers -> LDS between the barriers (the first version
// loaded them between the barriers: eight waves idle for an L2 round trip, five times per workgroup)
template <int STAGE>
__device__ __forceinline__ void ef_fetch(const float* __restrict__ wpack, f32x4 (&wr)[4]) {
  const int tid = threadIdx.x;
  const f32x4* s = reinterpret_cast<const f32x4*>(wpack + ENC_OFF_BLK(STAGE));
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (tid + q * EF_THREADS < ENC_BLK_FLOATS / 4) wr[q] = s[tid + q * EF_THREADS];
  if (STAGE == 0 || STAGE == 4) {
    const f32x4* s2 = reinterpret_cast<const f32x4*>(wpack + (STAGE == 0 ? ENC_OFF_FCPOS : ENC_OFF_FCC));
    if (tid < (STAGE == 0 ? 256 : 1056) / 4) wr[3] = s2[tid];
  }
}
static_assert(3 * EF_THREADS >= ENC_BLK_FLOATS / 4 && EF_THREADS >= 1056 / 4 && EF_THREADS >= EF_CAP, "ef_fetch covers a stage's weights with 3 + 1 registers per thread");

template <int STAGE>
__device__ __forceinline__ void ef_stage(float* __restrict__ lds, int* __restrict__ lsm, const float* __restrict__ wpack,
                                         const float* __restrict__ cloud, const int* __restrict__ order, long long bT,
                                         f32x16 (&net)[EF_NT], const EfTile (&ts)[EF_NT], f32x4 (&wr)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5;
  __syncthreads();                      // every wave is done with the previous stage: its weights, its pooled maxima are final
  f32x16 xhi[EF_NT];
  if (STAGE > 0) {
#pragma unroll
    for (int k = 0; k < EF_NT; ++k) {
      if (!ts[k].any) continue;
      const int4* sp = reinterpret_cast<const int4*>(lsm + ts[k].seg_l * EF_ROW + 4 * hi);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int4 q = sp[2 * g];
        xhi[k][4 * g + 0] = fkey_inv(q.x); xhi[k][4 * g + 1] = fkey_inv(q.y);
        xhi[k][4 * g + 2] = fkey_inv(q.z); xhi[k][4 * g + 3] = fkey_inv(q.w);
      }
    }
  }
  {
    f32x4* d = reinterpret_cast<f32x4*>(lds);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (tid + q * EF_THREADS < ENC_BLK_FLOATS / 4) d[tid + q * EF_THREADS] = wr[q];
    if ((STAGE == 0 || STAGE == 4) && tid < (STAGE == 0 ? 256 : 1056) / 4) reinterpret_cast<f32x4*>(lds + ENC_BLK_FLOATS)[tid] = wr[3];
  }
  __syncthreads();                      // the pooled maxima of the previous stage are in registers: their rows can be re-armed
  if (STAGE < 4) ef_fetch<STAGE < 4 ? STAGE + 1 : 4>(wpack, wr);
  if (STAGE < 4) {
    int4* r = reinterpret_cast<int4*>(lsm + tid * EF_ROW);    // EF_CAP rows, one per thread (of the first EF_CAP)
    const int e = (int)0x80808080;
    if (tid < EF_CAP) {
#pragma unroll
      for (int g = 0; g < 8; ++g) r[g] = int4{e, e, e, e};
    }
    __syncthreads();
  }
  const f32x4* L = reinterpret_cast<const f32x4*>(lds) + lane;
#pragma unroll
  for (int k = 0; k < EF_NT; ++k) {
    const EfTile& t = ts[k];
    if (!t.any) continue;               // an empty tile only keeps the barriers
    f32x16 xlo, xh;
    if (STAGE == 0) {
      const float* p = cloud + (bT + order[t.i]) * 3;
      const float hx = p[0] * 0.5f, hy = p[1] * 0.5f, hz = p[2] * 0.5f;  // vqdif.py:36 Xbd/2
      const float* fp = lds + ENC_BLK_FLOATS;
#pragma unroll
      for (int q = 0; q < 16; ++q) { xlo[q] = 0.0f; xh[q] = 0.0f; }
      xlo = MFMA(fp[lane], hi ? hy : hx, xlo);
      xlo = MFMA(fp[64 + lane], hi ? 1.0f : hz, xlo);
      xh = MFMA(fp[128 + lane], hi ? hy : hx, xh);
      xh = MFMA(fp[192 + lane], hi ? 1.0f : hz, xh);
    } else {
      xlo = net[k]; xh = xhi[k];
    }
    f32x16 h, o;
    bias_init(lds + 5 * 1024, hi, h);
    layer32<true>(L + 0 * 256, xlo, h);
    layer32<true>(L + 1 * 256, xh, h);
    bias_init(lds + 5 * 1024 + 32, hi, o);
    layer32<false>(L + 2 * 256, xlo, o);
    layer32<false>(L + 3 * 256, xh, o);
    layer32<true>(L + 4 * 256, h, o);
    if (STAGE < 4) {
      net[k] = o;
      if (t.valid) {           // local max pool (enc.py:95-112)
        int* sm = lsm + t.seg_l * EF_ROW + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicMax(sm + 8 * g + j, fkey(o[4 * g + j]));
      }
    } else {
      // c = fc_c(net) (enc.py:133), kept for the mean below
      f32x16 c;
      bias_init(lds + ENC_BLK_FLOATS + 1024, hi, c);
      layer32<false>(reinterpret_cast<const f32x4*>(lds + ENC_BLK_FLOATS) + lane, o, c);
      net[k] = c;
    }
  }
}

__global__ __launch_bounds__(EF_THREADS) void enc_fused_kernel(const float* __restrict__ cloud, const int* __restrict__ scell,
                                                               const int* __restrict__ start, const int* __restrict__ cend,
                                                               const int* __restrict__ order, long long* __restrict__ csum,
                                                               int* __restrict__ ccount, const float* __restrict__ wpack,
                                                               const int* __restrict__ flag, int T) {
  if (flag[1 + blockIdx.y]) return;     // a cell of this shape holds more than EF_LIMIT points: the staged kernels take the shape
  f32x4 wr[4];
  ef_fetch<0>(wpack, wr);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int* lsm = reinterpret_cast<int*>(lds + EF_W_FLOATS);       // [EF_CAP][EF_ROW] pooled maxima (ordered-int keys)
  __shared__ int s_rng[2];
  __shared__ int s_cnt[EF_CAP / 2];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pl = lane & 31, hi = lane >> 5;
  constexpr long long NC = (long long)ENC_G * ENC_G * ENC_G;
  const int* sc = scell + (long long)b * T;
  const int* st = start + b * NC;
  if (tid == 0) {
    int p0 = blockIdx.x * EF_NOM, p1 = min(T, p0 + EF_NOM);
    if (p0 < T && p0 > 0) { const int c = sc[p0]; if (st[c] < p0) p0 = cend[b * NC + c]; }   // the tail of a run begun before: its owner's
    if (p1 < T) { const int c = sc[p1]; if (st[c] < p1) p1 = cend[b * NC + c]; }             // the run that straddles the end: finished here
    s_rng[0] = p0; s_rng[1] = max(p0, p1);
  }
  __syncthreads();
  const int p0 = s_rng[0], n = s_rng[1] - p0;
  if (n <= 0) return;
  EfTile ts[EF_NT];
#pragma unroll
  for (int k = 0; k < EF_NT; ++k) {
    const int tl = (wave * EF_NT + k) * 32;       // the tile's first point, relative to p0
    EfTile& t = ts[k];
    t.valid = tl + pl < n;
    t.any = __builtin_amdgcn_readfirstlane(tl) < n;
    const int pos = p0 + (t.valid ? tl + pl : n - 1);
    t.i = b * T + pos;
    t.seg_l = st[sc[pos]] - p0;                   // sorted position of the cell's first point
  }
  f32x16 net[EF_NT];
  const long long bT = (long long)b * T;
  ef_stage<0>(lds, lsm, wpack, cloud, order, bT, net, ts, wr);
  ef_stage<1>(lds, lsm, wpack, cloud, order, bT, net, ts, wr);
  ef_stage<2>(lds, lsm, wpack, cloud, order, bT, net, ts, wr);
  ef_stage<3>(lds, lsm, wpack, cloud, order, bT, net, ts, wr);
  ef_stage<4>(lds, lsm, wpack, cloud, order, bT, net, ts, wr);     // net = c = fc_c(net)
  // scatter_mean numerator (enc.py:70-74): per-cell sums in 2^-32 fixed point (associative: any order gives the same bits) and
  // point counts, accumulated with LDS atomics in the pooled-maxima region - as 64-bit words it holds half of the rows, so the
  // cells that start in the first and in the second 256 positions take turns - and written out as whole rows: the fused form
  // needs neither global atomics nor zeroed sums.
  unsigned long long* lsum = reinterpret_cast<unsigned long long*>(lsm);      // [EF_CAP / 2][EF_ROW]
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half * (EF_CAP / 2) >= n) break;
    __syncthreads();                    // the region's previous users are done (pooled maxima read at the top of stage 4; the other half's rows written out)
    {
      int4* r = reinterpret_cast<int4*>(lsm + tid * EF_ROW);     // as ints: the whole region, i.e. all EF_CAP / 2 rows of 64-bit words
      if (tid < EF_CAP) {
#pragma unroll
        for (int g = 0; g < EF_ROW / 4; ++g) r[g] = int4{0, 0, 0, 0};
      }
      if (tid < EF_CAP / 2) s_cnt[tid] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EF_NT; ++k) {
      const EfTile& t = ts[k];
      if (t.valid && t.seg_l / (EF_CAP / 2) == half) {
        const int row = t.seg_l & (EF_CAP / 2 - 1);
        unsigned long long* sp = lsum + row * EF_ROW + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) atomicAdd(sp + 8 * g + j, (unsigned long long)__double2ll_rn((double)net[k][4 * g + j] * 4294967296.0));
        if (hi == 0) atomicAdd(s_cnt + row, 1);
      }
    }
    __syncthreads();
    {
      const int row = (tid >> 1) & (EF_CAP / 2 - 1), part = tid & 1, cnt = tid < EF_CAP ? s_cnt[row] : 0;
      if (cnt > 0) {                    // a cell starts at this position
        const long long seg = bT + p0 + half * (EF_CAP / 2) + row;
        const longlong2* src = reinterpret_cast<const longlong2*>(lsum + row * EF_ROW + part * 16);
        longlong2* dst = reinterpret_cast<longlong2*>(csum + seg * 32 + part * 16);
#pragma unroll
        for (int g = 0; g < 8; ++g) dst[g] = src[g];
        if (part == 0) ccount[seg] = cnt;
      }
    }
  }
}

// fill of a staged-form buffer (0x80808080: pooled maxima; 0: the per-cell sums its atomics add into), skipped when the fused kernel
// took every shape of the call
__global__ void enc_fill_kernel(int4* __restrict__ p, long long n4, const int* __restrict__ flag, int e) {
  if (flag && !*flag) return;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) p[i] = int4{e, e, e, e};
}

// E6: write the per-cell means into the dense channels-last (B,G,G,G,32) grid (pre-zeroed).
__global__ void enc_grid_mean_kernel(const int* __restrict__ cell, const int* __restrict__ rep,
                                     const long long* __restrict__ csum, const int* __restrict__ ccount,
                                     float* __restrict__ grid, int B, int T) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long i = gid >> 5;
  int ch = (int)(gid & 31);
  if (i >= (long long)B * T) return;
  int b = (int)(i / T), t = (int)(i - (long long)b * T);
  int c = cell[i];                                                 // sorted cell array: position t is a run start iff ...
  if (rep[(long long)b * ENC_G * ENC_G * ENC_G + c] != t) return;  // ... it is the cell's first sorted position
  double s = (double)csum[i * 32 + ch] * (1.0 / 4294967296.0);
  grid[((long long)b * ENC_G * ENC_G * ENC_G + c) * 32 + ch] = (float)(s / (double)ccount[i]);
}

// E6': the first Downsampler convolution (Conv3d(32 -> 64, k2, s2, no bias) + ReLU, updown.py:101-118) taken DIRECTLY from the
// per-cell sums - the dense 64^3 x 32 mean grid (33.5 MB per shape: a zero-fill, a scatter of ~3 000 occupied cells, and a full read
// by the convolution) is never materialised.  A parent voxel of the 32^3 output owns 8 child cells; the counting sort's start /
// end maps say which of them hold points (1-2 % do), the cell mean is formed exactly as enc_grid_mean_kernel forms it, and an
// empty child contributes exactly zero, as it does in the dense convolution.  One workgroup per (shape, zo, yo) row of 32 parents,
// wave w takes parents w, w+4, ...; lane = output channel; taps and input channels are summed in ascending order (fmaf): a fixed
// order, deterministic - the dense MFMA path sums the same products in another order (differences ~1e-7 relative).
__global__ __launch_bounds__(256, 4) void enc_down0_sparse_kernel(const int* __restrict__ start, const int* __restrict__ cend,
                                                               const long long* __restrict__ csum, const int* __restrict__ ccount,
                                                               const float* __restrict__ wt /*[8 taps][32 cin][64 cout]*/, float* __restrict__ y,
                                                               int T, int relu) {
  constexpr int G = ENC_G, GO = ENC_G / 2, NC = ENC_G * ENC_G * ENC_G;
  // one workgroup per (shape, zo) PLANE of 32 x 32 parents; wave w walks the rows yo = w, w + 4, ... with the next row's cell maps
  // in flight: 2 048 workgroups for 64 shapes, all resident at once (the first version - one workgroup per row - spent 32 dispatch
  // rounds on workgroups whose only work was one dependent load and 2 KB of zeros)
  const int b = blockIdx.y, zo = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int st[4], en[4], stn[4], enn[4];
  auto load_row = [&](int yo, int (&s_)[4], int (&e_)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // the four child rows (dz, dy) of parent row yo: 64 cells each, lane = child x
      const long long c = (long long)b * NC + ((long long)(2 * zo + (r >> 1)) * G + (2 * yo + (r & 1))) * G + lane;
      s_[r] = start[c]; e_[r] = cend[c];
    }
  };
  load_row(wave, st, en);
  for (int yo = wave; yo < GO; yo += 4) {
    if (yo + 4 < GO) load_row(yo + 4, stn, enn);
    unsigned long long occ[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) occ[r] = __ballot(en[r] != st[r]);
    float* yrow = y + ((((long long)b * GO + zo) * GO + yo) * GO) * 64;
    for (int xo = 0; xo < GO; ++xo) {
      float acc = 0.f;
      const unsigned bits = (unsigned)((occ[0] >> (2 * xo)) & 3) | (unsigned)(((occ[1] >> (2 * xo)) & 3) << 2) |
                            (unsigned)(((occ[2] >> (2 * xo)) & 3) << 4) | (unsigned)(((occ[3] >> (2 * xo)) & 3) << 6);
      if (bits) {                                   // wave-uniform: most parents have no occupied child
        // the sums of all 8 child slots are requested UNCONDITIONALLY (an empty slot reads the shape's first segment and is zeroed
        // afterwards): sixteen independent loads in one round trip - with the loads inside per-child branches every child cost two
        // dependent round trips of its own (26 us per occupied parent, 0.42 ms per 64 shapes)
        long long cs[8];
        int cc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int r = c >> 1, dx = c & 1;
          const int s0 = __shfl(st[r], 2 * xo + dx, 64);
          const long long seg = (long long)b * T + (((bits >> c) & 1) ? s0 : 0);      // sorted position of the cell's first point names its sums
          cs[c] = csum[seg * 32 + (lane & 31)];
          cc[c] = ccount[seg];
        }
        float mean[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {     // cell mean of channel (lane & 31), formed as enc_grid_mean_kernel does
          const double sm = (double)cs[c] * (1.0 / 4294967296.0);
          mean[c] = ((bits >> c) & 1) ? (float)(sm / (double)cc[c]) : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (!((bits >> c) & 1)) continue;
          const int r = c >> 1, dx = c & 1;
          const int tap = ((r >> 1) * 2 + (r & 1)) * 2 + dx;        // (dz, dy, dx), the conv kernel's tap order
          // weights transposed to [tap][cin][cout]: one coalesced 256-byte load per input channel, all 32 in flight together.
          // The offset is laundered through an empty asm: the weights do not depend on the parent, and hoisting all 8 x 32 of
          // them out of the parent loop (what the optimiser does otherwise) costs 256 registers and the kernel's occupancy
          int woff = tap * 32 * 64;
          asm volatile("" : "+s"(woff));
          const float* wp = wt + woff + lane;
          float wv[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) wv[k] = wp[k * 64];
#pragma unroll
          for (int k = 0; k < 32; ++k) acc = fmaf(wv[k], __shfl(mean[c], k, 64), acc);
        }
      }
      yrow[(long long)xo * 64 + lane] = relu ? fmaxf(acc, 0.f) : acc;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[r] = stn[r]; en[r] = enn[r]; }
  }
}

// [8 taps][64 cout][32 cin] (sfmi_conv_pack_weight layout) -> [8][32][64] for enc_down0_sparse_kernel
__global__ void enc_down0_wt_kernel(const float* __restrict__ w, float* __restrict__ wt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // over 8 * 64 * 32
  if (i >= 8 * 64 * 32) return;
  const int ci = i & 31, co = (i >> 5) & 63, tap = i >> 11;
  wt[(tap * 32 + ci) * 64 + co] = w[i];
}

extern "C" {

size_t sfmi_enc_pack_floats(void) { return ENC_PACK_FLOATS; }

static void pack_frag(const float* W, int ld, int col0, float* out) {  // W[co][col0 + k], k<32
  for (int g = 0; g < 4; ++g)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) out[(g * 64 + l) * 4 + j] = W[(l & 31) * ld + col0 + 8 * g + 4 * (l >> 5) + j];
}

// Host packer. blocks: 5x {fc_0.weight(32,64), fc_0.bias, fc_1.weight(32,32), fc_1.bias, shortcut.weight(32,64)}
int sfmi_enc_pack_weights(const float* fc_pos_w /*64x3*/, const float* fc_pos_b /*64*/, const float* fc0_w,
                          const float* fc0_b, const float* fc1_w, const float* fc1_b, const float* sc_w,
                          const float* fc_c_w /*32x32*/, const float* fc_c_b, float* out) {
  if (!fc_pos_w || !out) return SFMI_EINVAL;
  for (int k = 0; k < 5; ++k) {
    float* o = out + ENC_OFF_BLK(k);
    pack_frag(fc0_w + k * 2048, 64, 0, o);
    pack_frag(fc0_w + k * 2048, 64, 32, o + 1024);
    pack_frag(sc_w + k * 2048, 64, 0, o + 2048);
    pack_frag(sc_w + k * 2048, 64, 32, o + 3072);
    pack_frag(fc1_w + k * 1024, 32, 0, o + 4096);
    for (int c = 0; c < 32; ++c) {
      o[5120 + c] = fc0_b[k * 32 + c];
      o[5152 + c] = fc1_b[k * 32 + c];
    }
  }
  float* fp = out + ENC_OFF_FCPOS;
  for (int half = 0; half < 2; ++half)
    for (int l = 0; l < 64; ++l) {
      int co = half * 32 + (l & 31), hi = l >> 5;
      fp[half * 128 + l] = fc_pos_w[co * 3 + (hi ? 1 : 0)];
      fp[half * 128 + 64 + l] = hi ? fc_pos_b[co] : fc_pos_w[co * 3 + 2];
    }
  pack_frag(fc_c_w, 32, 0, out + ENC_OFF_FCC);
  for (int c = 0; c < 32; ++c) out[ENC_OFF_FCC + 1024 + c] = fc_c_b[c];
  return SFMI_OK;
}

size_t sfmi_enc_workspace_bytes(int B, int T) {
  size_t bt = (size_t)B * T;
  // cell + order + sorted cell (3 x 4) + start map + cursor map + 2 net buffers + 2 segmax buffers + csum + ccount + scan chunk sums
  return bt * 12 + 2 * (size_t)B * ENC_G * ENC_G * ENC_G * 4 + 2 * bt * 128 + 2 * bt * 128 + bt * 256 + bt * 4 + 256 + (((size_t)B + 1) * 4 + 255) + 256 + (size_t)B * 64 * 4 + 65536 + 1024;
}

static int enc_pipeline(const float* cloud, const float* wpack, float* grid_cl, unsigned char* mask, int* cell_out, void* workspace, int B, int T,
                        int R, float* tap_stage1, float* tap_stage4c, const float* down_w, float* down_y, int down_relu, void* stream_);
// replaces LocalPoolPointnet.forward up to scatter_mean (enc.py:115-140 minus the Downsampler):
// cloud (B,T,3) -> dense channels-last mean grid (B,64,64,64,32) + latent occupancy mask (B,R,R,R) u8.
int sfmi_encode_points_f32(const float* cloud, const float* wpack, float* grid_cl, unsigned char* mask,
                           int* cell_out /*optional (B,T)*/, void* workspace, int B, int T, int R, void* stream_) {
  if (!grid_cl) return SFMI_EINVAL;
  return enc_pipeline(cloud, wpack, grid_cl, mask, cell_out, workspace, B, T, R, nullptr, nullptr, nullptr, nullptr, 0, stream_);
}
// the same encoder with the FIRST Downsampler convolution fused in (enc.py:115-140 + the first Conv3d(32 -> 64, k2, s2, no bias)
// (+ ReLU) of updown.py:101-118): cloud -> y (B,32,32,32,64) channels-last, taken from the per-cell sums without materialising the
// dense 64^3 x 32 mean grid.  w_down0: the convolution's weights as [8 taps (dz,dy,dx)][64][32] (the layout sfmi_conv3d_cl_f32 takes).
int sfmi_encode_points_down_f32(const float* cloud, const float* wpack, const float* w_down0, float* y, unsigned char* mask,
                                int* cell_out /*optional (B,T)*/, void* workspace, int B, int T, int R, int relu, void* stream_) {
  if (!w_down0 || !y) return SFMI_EINVAL;
  return enc_pipeline(cloud, wpack, nullptr, mask, cell_out, workspace, B, T, R, nullptr, nullptr, w_down0, y, relu, stream_);
}
// the same pipeline with per-point taps for stage-wise parity tests (caller's point order): tap_stage1 (B,T,32) = output of
// blocks[1] (after the first local max pool), tap_stage4c (B,T,64) = [output of blocks[4] | c = fc_c(net)] (enc.py:124-133)
int sfmi_encode_points_tap_f32(const float* cloud, const float* wpack, float* grid_cl, unsigned char* mask, int* cell_out, void* workspace,
                               int B, int T, int R, float* tap_stage1, float* tap_stage4c, void* stream_) {
  if (!grid_cl) return SFMI_EINVAL;
  return enc_pipeline(cloud, wpack, grid_cl, mask, cell_out, workspace, B, T, R, tap_stage1, tap_stage4c, nullptr, nullptr, 0, stream_);
}
static int enc_pipeline(const float* cloud, const float* wpack, float* grid_cl, unsigned char* mask, int* cell_out, void* workspace, int B, int T,
                        int R, float* tap_stage1, float* tap_stage4c, const float* down_w, float* down_y, int down_relu, void* stream_) {
  if (!cloud || !wpack || (!grid_cl && !down_y) || !mask || !workspace || B <= 0 || T <= 0 || R <= 0) return SFMI_EINVAL;
  hipStream_t st = (hipStream_t)stream_;
  const size_t bt = (size_t)B * T, nc = (size_t)B * ENC_G * ENC_G * ENC_G;
  char* w = (char*)workspace;
  int* cell = (int*)w; w += bt * 4;
  int* order = (int*)w; w += bt * 4;
  int* scell = (int*)w; w += bt * 4;
  int* start = (int*)w; w += nc * 4;
  int* cursor = (int*)w; w += nc * 4;
  float* net[2]; net[0] = (float*)w; w += bt * 128; net[1] = (float*)w; w += bt * 128;
  int* sm[2]; sm[0] = (int*)w; w += bt * 128; sm[1] = (int*)w; w += bt * 128;
  long long* csum = (long long*)w; w += bt * 256;
  const size_t cc_bytes = (bt * 4 + 255) & ~(size_t)255;
  int* ccount = (int*)w; w += cc_bytes;
  const size_t flag_bytes = (((size_t)B + 1) * 4 + 255) & ~(size_t)255;
  int* flag = (int*)w; w += flag_bytes;                  // [0] any shape / [1 + b] shape b: a cell holds more than EF_LIMIT points (zeroed with csum / ccount)
  int* chunk_sum = (int*)w; w += (size_t)B * 64 * 4;     // (B, 64) chunk totals of the cell-count scan
  w = (char*)(((uintptr_t)w + 255) & ~(uintptr_t)255);
  float* down_wt = (float*)w;    // [8][32][64] transposed copy of the first Downsampler convolution's weights (64 KB)
  hipMemsetAsync(start, 0, nc * 4, st);
  hipMemsetAsync(mask, 0, (size_t)B * R * R * R, st);
  // per-cell sums and counts: the fused kernel writes whole rows, only the staged kernels' atomics need them zeroed (conditional fill below)
  hipMemsetAsync(flag, 0, flag_bytes, st);
  if (grid_cl) hipMemsetAsync(grid_cl, 0, nc * 32 * 4, st);
  int nb = (int)((bt + 255) / 256);
  // the five stages in one launch (enc_fused_kernel) unless a debug tap wants the per-stage buffers or the knob says otherwise;
  // limit < 0 makes the scan raise the flag unconditionally
  const bool fused = g_sfmi_tune.enc_fused && !tap_stage1 && !tap_stage4c;
  // group the points of every shape by cell: histogram -> exclusive scan -> scatter (2 integer atomics per point)
  hipLaunchKernelGGL(enc_cells_kernel, dim3(nb), dim3(256), 0, st, cloud, cell, start, mask, B, T, R);
  hipLaunchKernelGGL(enc_scan_sums_kernel, dim3(ENC_NCH, B), dim3(1024), 0, st, start, chunk_sum);
  hipLaunchKernelGGL(enc_scan_apply_kernel, dim3(ENC_NCH, B), dim3(1024), 0, st, start, cursor, chunk_sum, flag, fused ? EF_LIMIT : -1);
  hipLaunchKernelGGL(enc_scatter_kernel, dim3(nb), dim3(256), 0, st, cell, cursor, order, scell, B, T);
  // (before the fused kernel: it writes rows of the shapes it takes)
  hipLaunchKernelGGL(enc_fill_kernel, dim3(2048), dim3(256), 0, st, (int4*)csum, (long long)((bt * 256 + cc_bytes) / 16), fused ? flag : nullptr, 0);
  if (fused) {
    constexpr size_t ef_lds = (size_t)(EF_W_FLOATS + EF_CAP * EF_ROW) * 4;   // 98.7 KB: one workgroup of eight waves per CU
    static const hipError_t attr = hipFuncSetAttribute((const void*)enc_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ef_lds);
    if (attr != hipSuccess) return SFMI_ELDS;
    hipLaunchKernelGGL(enc_fused_kernel, dim3((T + EF_NOM - 1) / EF_NOM, B), dim3(EF_THREADS), ef_lds, st, cloud, scell, start, cursor, order,
                       csum, ccount, wpack, flag, T);
  }
  // the staged form: one launch per stage (the pool between stages as a global dependency).  With the fused kernel ahead of them
  // these launches test the flag and return (4.5 us each, measured) unless it declined a shape.
  const int* rf = fused ? flag : nullptr;
  const long long tiles = (long long)(bt + 31) / 32;
  int grid = (int)((tiles + 3) / 4);
  if (grid > 2048) grid = 2048;
  const size_t lds0 = (ENC_BLK_FLOATS + 256) * 4, ldsk = ENC_BLK_FLOATS * 4, lds4 = (ENC_BLK_FLOATS + 1056) * 4;
  const long long n4 = (long long)bt * 8;       // int4s of a (B,T,32) pooled-maxima buffer
  const int e80 = (int)0x80808080;
  hipLaunchKernelGGL(enc_fill_kernel, dim3(2048), dim3(256), 0, st, (int4*)sm[0], n4, rf, e80);
  hipLaunchKernelGGL(enc_block_kernel<0>, dim3(grid), dim3(256), lds0, st, cloud, scell, start, order, nullptr, nullptr, net[0],
                     sm[0], nullptr, nullptr, wpack, B, T, rf);
  hipLaunchKernelGGL(enc_fill_kernel, dim3(2048), dim3(256), 0, st, (int4*)sm[1], n4, rf, e80);
  hipLaunchKernelGGL(enc_block_kernel<1>, dim3(grid), dim3(256), ldsk, st, cloud, scell, start, order, net[0], sm[0], net[1],
                     sm[1], nullptr, nullptr, wpack, B, T, rf);
  if (tap_stage1)
    hipLaunchKernelGGL(enc_unsort_kernel, dim3((unsigned)((bt * 32 + 255) / 256)), dim3(256), 0, st, net[1], order, tap_stage1, B, T, 32);
  hipLaunchKernelGGL(enc_fill_kernel, dim3(2048), dim3(256), 0, st, (int4*)sm[0], n4, rf, e80);
  hipLaunchKernelGGL(enc_block_kernel<2>, dim3(grid), dim3(256), ldsk, st, cloud, scell, start, order, net[1], sm[1], net[0],
                     sm[0], nullptr, nullptr, wpack, B, T, rf);
  hipLaunchKernelGGL(enc_fill_kernel, dim3(2048), dim3(256), 0, st, (int4*)sm[1], n4, rf, e80);
  hipLaunchKernelGGL(enc_block_kernel<3>, dim3(grid), dim3(256), ldsk, st, cloud, scell, start, order, net[0], sm[0], net[1],
                     sm[1], nullptr, nullptr, wpack, B, T, rf);
  hipLaunchKernelGGL(enc_block_kernel<4>, dim3(grid), dim3(256), lds4, st, cloud, scell, start, order, net[1], sm[1], tap_stage4c,
                     nullptr, csum, ccount, wpack, B, T, rf);
  long long nthr = (long long)bt * 32;
  if (grid_cl)
    hipLaunchKernelGGL(enc_grid_mean_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, scell, start, csum,
                       ccount, grid_cl, B, T);
  if (down_y) {   // `cursor` has been advanced to each cell's END by the scatter: cursor != start <=> the cell holds points
    hipLaunchKernelGGL(enc_down0_wt_kernel, dim3(64), dim3(256), 0, st, down_w, down_wt);
    hipLaunchKernelGGL(enc_down0_sparse_kernel, dim3(ENC_G / 2, B), dim3(256), 0, st, start, cursor, csum, ccount, down_wt, down_y,
                       T, down_relu);
  }
  if (cell_out) hipMemcpyAsync(cell_out, cell, bt * 4, hipMemcpyDeviceToDevice, st);
  SFMI_CHECK_LAUNCH();
  return SFMI_OK;
}

}  // extern "C"
