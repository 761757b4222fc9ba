// Bare read stream with the decode attention's access pattern (csrc/gpt.hip:attn_decode_item), for the package-power
// calibration VERDICT r4 item 1 asks for: what does pulling the AR loop's 31.4 GB of f32 K/V per step cost in watts when the
// kernel does NOTHING with the bytes but add them up?
//
// An "item" is one (row, head) of one layer: 2 * L keys/values x 64 floats, contiguous (K rows then V rows).  A workgroup of NWV
// waves walks an item in batches of NWV * 4 * U keys; one wave-load = 4 keys x 256 B = 1 KiB contiguous (lane = (key lane>>4,
// float4 lane&15)), exactly the product kernel's f32x4 pattern.  Policies: 0 = `global_load_dwordx4 ... nt` (the product),
// 1 = default policy, 2 = buffer load sc1 (L1 bypass), 3 = buffer load default (separates buffer-vs-global from sc1).
//
// Built as a shared library (hipcc --offload-arch=gfx950 -shared -fPIC) and driven by tools/stream_power.py under the same
// hwmon power probe as tools/ar_sweep.py.  Timing-only: the sums are written only to keep the loads alive.
#include "../../shapeformer_amd/csrc/sfmi_common.h"      // f32x4, row16_sum (DPP), wave_max

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NWV, int U, int POL>
__global__ __launch_bounds__(64 * NWV) void stream_kernel(const float* __restrict__ buf, float* __restrict__ out, int item0, int nitems, int L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c4 = lane & 15, kk = lane >> 4;
  constexpr int KB = NWV * 4 * U;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const float* base = buf + (long long)(item0 + it) * 2 * L * 64;
    const int n = 2 * L;      // keys then values: one contiguous run of 2 L rows of 256 B
    if constexpr (POL >= 2) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n * 256, 0x00020000);
      for (int i0 = 0; i0 < n; i0 += KB) {
        u32x4 kf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * (NWV * 4) + wave * 4 + kk;      // out-of-range rows read zero (hardware bounds check)
          kf[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 256 + 16 * c4, 0, POL == 2 ? 16 : 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc[0] += __uint_as_float(kf[u][0]); acc[1] += __uint_as_float(kf[u][1]);
          acc[2] += __uint_as_float(kf[u][2]); acc[3] += __uint_as_float(kf[u][3]);
        }
      }
    } else {
      for (int i0 = 0; i0 < n; i0 += KB) {
        f32x4 kf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
          kf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (i < n) {
            const f32x4* p = reinterpret_cast<const f32x4*>(base + (long long)i * 64 + 4 * c4);
            kf[u] = POL == 0 ? __builtin_nontemporal_load(p) : *p;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = acc + kf[u];
      }
    }
  }
  const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (s == 1.2345678e-30f) out[blockIdx.x] = s;      // never true on random data; keeps the loads
}

// ------------------------------------------------------------------------------------------------------------------------------
// The decode attention's structure (csrc/gpt.hip:attn_decode_item) rebuilt INGREDIENT BY INGREDIENT on top of the bare stream, to see
// which one costs the 15 % between the bare stream (7.15 TB/s) and the attention family alone (6.08 TB/s).  Timing only.
//   FLAGS bit 0 (1):  scores: 4-float dot product per lane + 16-lane DPP sum + running max
//         bit 1 (2):  scores written to / read back from LDS (s.sc)
//         bit 2 (4):  wave_max + cross-wave max through LDS + the mid-kernel barrier (two-pass softmax)
//         bit 3 (8):  exp(s - max) and acc += p * v   (else: acc += v)
//         bit 4 (16): prologue: q / k / v of the new token through LDS + barrier, new K / V row appended to the cache
//         bit 5 (32): epilogue: cross-lane shuffles, per-wave partials through LDS, barrier, 64-float output row
// FLAGS = 0 is the two-loop form of the stream (keys, then values, first value batch requested before the value loop).
// ------------------------------------------------------------------------------------------------------------------------------
template <int NWV, int U, int FLAGS>
__global__ __launch_bounds__(64 * NWV, 8) void attn_like_kernel(float* __restrict__ buf, const float* __restrict__ qkv, float* __restrict__ out, int item0,
                                                                 int nitems, int L) {
  __shared__ __attribute__((aligned(16))) float qs[64], kn[64], vn[64], sc[1024], red[2 * NWV], yacc[NWV][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c4 = lane & 15, kk = lane >> 4;
  constexpr int KB = NWV * 4 * U;
  const int it = blockIdx.x;
  if (it >= nitems) return;
  float* Kb = buf + (long long)(item0 + it) * 2 * L * 64;
  float* Vb = Kb + (long long)L * 64;
  const int t = L - 1;      // position being processed: keys 0 .. t-1 come from the cache, key t is the new token
  auto load = [&](const float* base, int i0, f32x4 (&kf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      kf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < t) kf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (long long)i * 64 + 4 * c4));
    }
  };
  f32x4 kf0[U], vf0[U];
  load(Kb, 0, kf0);
  if (FLAGS & 16) {
    if (tid < 64) {
      const float q = qkv[(item0 + it) % 4096 * 192 + tid], k = qkv[(item0 + it) % 4096 * 192 + 64 + tid], v = qkv[(item0 + it) % 4096 * 192 + 128 + tid];
      qs[tid] = q * 0.125f; kn[tid] = k; vn[tid] = v;
      Kb[(long long)t * 64 + tid] = k;
      Vb[(long long)t * 64 + tid] = v;
    }
    __syncthreads();
  }
  f32x4 qf = {0.5f, -0.25f, 0.125f, 1.0f};
  if (FLAGS & 16) qf = *reinterpret_cast<const f32x4*>(qs + 4 * c4);
  float lmax = -INFINITY;
  f32x4 junk = {0.f, 0.f, 0.f, 0.f};
  auto score = [&](int i0, f32x4 (&kf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      if (FLAGS & 1) {
        if ((FLAGS & 16) && i == t) kf[u] = *reinterpret_cast<const f32x4*>(kn + 4 * c4);
        float d = (qf[0] * kf[u][0] + qf[1] * kf[u][1]) + (qf[2] * kf[u][2] + qf[3] * kf[u][3]);
        d = row16_sum(d);
        if (i <= t) {
          if ((FLAGS & 2) && c4 == 0) sc[i] = d;
          lmax = fmaxf(lmax, d);
        }
      } else {
        junk = junk + kf[u];
      }
    }
  };
  score(0, kf0);
  for (int i0 = KB; i0 <= t; i0 += KB) {
    f32x4 kf[U];
    load(Kb, i0, kf);
    score(i0, kf);
  }
  load(Vb, 0, vf0);
  float gmax = lmax;
  if (FLAGS & 4) {
    lmax = wave_max(lmax);
    if (lane == 0) red[wave] = lmax;
    __syncthreads();
    gmax = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) gmax = fmaxf(gmax, red[w]);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float ls = 0.f;
  auto accum = [&](int i0, f32x4 (&vf)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * (NWV * 4) + wave * 4 + kk;
      if ((FLAGS & 16) && i == t) vf[u] = *reinterpret_cast<const f32x4*>(vn + 4 * c4);
      if (FLAGS & 8) {
        const float sv = (FLAGS & 2) ? sc[min(i, 1023)] : lmax;
        const float pr = i <= t ? __expf(sv - gmax) : 0.f;
        acc = acc + vf[u] * pr;
        ls += pr;
      } else {
        acc = acc + vf[u];
      }
    }
  };
  accum(0, vf0);
  for (int i0 = KB; i0 <= t; i0 += KB) {
    f32x4 vf[U];
    load(Vb, i0, vf);
    accum(i0, vf);
  }
  acc = acc + junk;
  if (FLAGS & 32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] += __shfl_xor(acc[e], 16, 64); acc[e] += __shfl_xor(acc[e], 32, 64); }
    ls += __shfl_xor(ls, 16, 64); ls += __shfl_xor(ls, 32, 64);
    if (kk == 0) *reinterpret_cast<f32x4*>(&yacc[wave][4 * c4]) = acc;
    if (lane == 0) red[NWV + wave] = ls;
    __syncthreads();
    if (tid < 64) {
      float o = 0.f, l = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) { o += yacc[w][tid]; l += red[NWV + w]; }
      out[(long long)(it % 4096) * 64 + tid] = o / (l + 1.0f);
    }
  } else {
    const float x = (acc[0] + acc[1]) + (acc[2] + acc[3]) + ls + gmax;
    if (x == 1.2345678e-30f) out[it % 4096] = x;
  }
}

extern "C" {

// attention-like variants: flags as documented above; launches one grid of `nitems` workgroups over items [item0, item0 + nitems)
int sp_launch_attn(int flags, float* buf, const float* qkv, float* out, int item0, int nitems, int L, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (L > 1024) return -2;
#define AL(F_) case F_: hipLaunchKernelGGL((attn_like_kernel<16, 4, F_>), dim3(nitems), dim3(1024), 0, st, buf, qkv, out, item0, nitems, L); break;
  switch (flags) {
    AL(0) AL(1) AL(3) AL(5) AL(7) AL(15) AL(31) AL(47) AL(63) AL(59) AL(55)
    default: return -1;
  }
#undef AL
  return (int)hipGetLastError();
}

// variant = waves * 1000 + U * 10 + policy, e.g. 16040 = the product's launch shape.  Launches ONE grid over items
// [item0, item0 + nitems) with `blocks` workgroups (blocks >= nitems: one item per workgroup, as the product launches).
int sp_launch(int variant, const float* buf, float* out, int item0, int nitems, int L, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (blocks > nitems) blocks = nitems;
#define SP(W_, U_, P_) case W_ * 1000 + U_ * 10 + P_: \
    hipLaunchKernelGGL((stream_kernel<W_, U_, P_>), dim3(blocks), dim3(64 * W_), 0, st, buf, out, item0, nitems, L); break;
  switch (variant) {
    SP(16, 4, 0) SP(16, 4, 1) SP(16, 4, 2) SP(16, 4, 3)
    SP(16, 2, 0) SP(16, 8, 0)
    SP(8, 8, 0) SP(8, 4, 0) SP(8, 8, 1) SP(8, 8, 2)
    SP(4, 16, 0) SP(4, 8, 0) SP(4, 16, 1) SP(4, 16, 2)
    default: return -1;
  }
#undef SP
  return (int)hipGetLastError();
}

}  // extern "C"
