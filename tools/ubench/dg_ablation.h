// Timing-only ablation policies for csrc/gpt.hip:dgemm_kernel (micro-benchmarks only; results are wrong by construction).
// Include AFTER "../../shapeformer_amd/csrc/gpt.hip".  Build-time switches of the run_*.sh scripts:
//   -DDG_NO_XLOAD     activation operand always fragment 0 (no activation traffic)
//   -DDG_NO_WLOAD     weight operand is a constant (no weight traffic)
//   -DDG_NO_MFMA      the MFMA is replaced by one FMA
//   -DDG_STATS_IF_LN  LayerNorm statistics only in the launches that fuse a LayerNorm;  -DDG_NO_STATS  never
//   -DDG_SKIP_EPI     main loop only
//   -DDG_FORCE_NW=n / -DDG_FORCE_UN=n   set the dgemm_nw / dgemm_un knobs (csrc/sfmi_common.h) before main runs
#pragma once
struct DgAblate {
#ifdef DG_STATS_IF_LN
  static constexpr bool kStatsOnlyIfLn = true;
#else
  static constexpr bool kStatsOnlyIfLn = false;
#endif
#ifdef DG_NO_STATS
  static constexpr bool kNoStats = true;
#else
  static constexpr bool kNoStats = false;
#endif
#ifdef DG_SKIP_EPI
  static constexpr bool kSkipEpilogue = true;
#else
  static constexpr bool kSkipEpilogue = false;
#endif
  static __device__ __forceinline__ f32x4 wload(const f32x4* p) {
#ifdef DG_NO_WLOAD
    return f32x4{1.f, 2.f, 3.f, 4.f};
#else
    return *p;
#endif
  }
  static __device__ __forceinline__ int xidx(int i) {
#ifdef DG_NO_XLOAD
    return 0;
#else
    return i;
#endif
  }
  static __device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
#ifdef DG_NO_MFMA
    c[0] += a * b; return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
  }
};
static inline int dg_call(const float* x, const float* Wp16, const float* c1, const float* c2, const float* resid, float* out, int M, int N,
                          int K, int ldo, int ln, int act, int out_packed, int S, float* slab, int* cnt, void* stream) {
  return decode_gemm_launch<DgAblate>(x, Wp16, c1, c2, resid, out, M, N, K, ldo, ln, act, out_packed, S, slab, cnt, nullptr, nullptr, stream);
}
namespace {
struct DgKnobs {
  DgKnobs() {
#ifdef DG_FORCE_NW
    g_sfmi_tune.dgemm_nw = DG_FORCE_NW;
#endif
#ifdef DG_FORCE_UN
    g_sfmi_tune.dgemm_un = DG_FORCE_UN;
#endif
  }
} dg_knobs_init;
}
