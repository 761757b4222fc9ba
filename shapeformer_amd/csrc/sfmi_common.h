// Shared device helpers for libsfmi (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SFMI_OK 0
#define SFMI_EINVAL (-1)
#define SFMI_ELAUNCH (-2)
#define SFMI_ELDS (-4)   /* the device refused the dynamic-LDS size a kernel needs */

#define SFMI_CHECK_LAUNCH()                       \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

// reference constants: vqdif/common.py:260-276 (padding 0.1, "10e-4" == 1e-3)
#define SFMI_NORM_DIV 1.101f
#define SFMI_NORM_HI 0.999f

// a2: normalize_3d_coordinate on one coordinate (p already in [-.5,.5])
__device__ __forceinline__ float sfmi_normalize(float p) {
  float u = __fdiv_rn(p, SFMI_NORM_DIV) + 0.5f;
  u = (u >= 1.0f) ? SFMI_NORM_HI : u;
  u = (u < 0.0f) ? 0.0f : u;
  return u;
}

// Sum / max over each aligned group of 16 lanes (a DPP "row"), result in every lane of the group: the xor-1 / 2 / 4 / 8 butterfly with
// the exchanges done by DPP operand modifiers (quad_perm, row_half_mirror, row_mirror) inside the adds instead of four ds_bpermute
// round trips through the LDS crossbar.  Bit-identical to the __shfl_xor butterfly: after the quad steps all four lanes of a quad hold
// the same value, so "the lane 7-i / 15-i away" and "the lane i^4 / i^8 away" deliver the same addend, and a + b == b + a.
__device__ __forceinline__ float dpp_f(float v, const int ctrl_sel) {
  // ctrl_sel: 0 quad_perm[1,0,3,2], 1 quad_perm[2,3,0,1], 2 row_half_mirror, 3 row_mirror (compile-time constants below)
  int r;
  const int x = __float_as_int(v);
  if (ctrl_sel == 0) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);
  else if (ctrl_sel == 1) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);
  else if (ctrl_sel == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false);
  else r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false);
  return __int_as_float(r);
}
__device__ __forceinline__ float row16_sum(float d) {
  d += dpp_f(d, 0); d += dpp_f(d, 1); d += dpp_f(d, 2); d += dpp_f(d, 3);
  return d;
}
__device__ __forceinline__ float row16_max(float d) {
  d = fmaxf(d, dpp_f(d, 0)); d = fmaxf(d, dpp_f(d, 1)); d = fmaxf(d, dpp_f(d, 2)); d = fmaxf(d, dpp_f(d, 3));
  return d;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// counter-hash uniform in [0,1): element idx of weights.hash_unit(key) with seed = fnv1a32(key) (murmur3 finalizer).  Shared
// by the sampler's inverse-CDF draws and the training dropout masks, so the CPU oracle reproduces both exactly.
__device__ __forceinline__ float sfmi_hash_unit(unsigned seed, unsigned idx) {
  unsigned h = idx * 0x9E3779B1u + seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
// nn.Dropout(p) multiplier of element idx: 0 with probability p, else 1/(1-p)  (mingpt.py:62-63,85,90,105,218)
__device__ __forceinline__ float sfmi_dropout_mul(unsigned seed, unsigned idx, float p, float inv_keep) {
  return sfmi_hash_unit(seed, idx) < p ? 0.0f : inv_keep;
}

// Launch-shape tuning knobs (performance only: none changes a result bit unless its comment says so).  Set through
// sfmi_tune_set(name, value) (csrc/capi.hip) by bench.py / tools/ar_sweep.py; read at LAUNCH time, i.e. baked into a captured
// hipGraph (re-capture after changing one).  Defaults = the product configuration.
//   attn_blocks : 0 = one workgroup per (row, head) item of the decode attention; n > 0 = persistent grid of n workgroups
//   attn_unroll : float4 loads in flight per lane (2, 4 or 8; 8 or 16 with attn_waves = 4)
//   attn_waves  : 16 or 8 waves per workgroup (NOT bit-identical to each other: different summation order); 4 = the light-occupancy
//                 experiment of round 5 (a quarter of a CU's wave slots per two workgroups, profiles/r05_stream_power.md)
//   attn_lds_pad: extra dynamic LDS bytes per attention workgroup (caps resident workgroups per CU)
//   sdf_blocks  : cap of the SDF-query kernel's persistent grid (default 512 = two workgroups per CU; 256 leaves half of every
//                 CU's registers free - the background-decode experiment of profiles/r03_ar_overlap.md)
//   dgemm_nt2   : decode GEMM with two n-tiles per wave and row groups of <= 3 row tiles (bit-identical to the one-tile form):
//                 0 = never, 1 = when the row tiles divide by 3 (default: 48 / 96 / 144 / 192 rows), 2 = from 48 rows on
//   dgemm_nw    : k-parts (waves) per decode-GEMM workgroup: 0 = by shape (default), 4 / 8 / 16 where the K-slice allows it
//   dgemm_un    : cap on the k16-steps of loads in flight per wave: 0 = by shape (8 / 4 / 2 / 2 / 2 / 1 for 1 .. 6 row tiles); can only lower it
// sfmi_tune_generation() counts successful sfmi_tune_set calls: callers that cache captured hipGraphs key them on it.
//   conv_xreuse : stride-1 k2 / k3 convolutions stage each input row once per (dz, dy) and reuse it for the taps along x: 1 = with 32 / 64
//                 output channels per tile (round 4 form); 2 (default) = also UNet3D's 128-multiple layers, on 128 x 64 tiles whose MFMA
//                 chain is folded into a second accumulator every 384 products (3 x less rounding noise per layer than the plain
//                 chain); 3 = those layers on 128 x 128 tiles with swizzled 16-float LDS rows (5 % faster, plain chain); 0 = re-stage
//                 per tap (round 1-3 form).  The forms are NOT bit-identical to each other (summation order: fp32 rounding only)
//   sk_grid     : workgroups of the work-balanced training GEMM (csrc/sgemm_sk.hip): 512 (default: two per CU) / 256 / 768 / 1024.  NOT
//                 bit-identical to each other (a tile's K range is cut at other places: fp32 rounding only)
//   sk_tile     : its workgroup tile: 0 = by output size (default), 1 = 64 x 64, 2 = 128 x 128 (same remark)
//   sk_loop     : chunk loop of its 64 x 64 form: 0 = store / barrier / load / read / MFMA (default), 1 = pipelined + interleaved (bit-identical)
//   sk_stagger  : experiment: the second resident workgroup of a CU starts `sk_stagger` x 256 cycles late (0 = off; bit-identical)
//   enc_fused   : 1 (default) = the five per-point stages of the local-pool encoder in one launch (run-aligned workgroups, csrc/encoder.hip
//                 enc_fused_kernel); 0 = one launch per stage (round 1-4 form).  Bit-identical to each other.
//   dgemm_prio  : wave priority (s_setprio 0..3) of the decode GEMM's waves: beside the KV stream of another chain the two families
//                 share each SIMD's issue slots; 0 = the hardware default (round 6 A/B, profiles/r06_overlap.md).  Scheduling only.
struct SfmiTune { int attn_blocks, attn_unroll, attn_waves, attn_lds_pad, sdf_blocks, dgemm_nt2, dgemm_nw, dgemm_un, conv_xreuse, sk_grid, sk_tile, sk_loop, sk_stagger, enc_fused, dgemm_prio; };
extern SfmiTune g_sfmi_tune;
