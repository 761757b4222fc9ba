// LDS-staged decode GEMM (csrc/dgemm_lds.hip) against the n-tile-per-workgroup kernel (csrc/gpt.hip): numerics on random
// operands, then both in the same hipGraph chain of 48 launches over 24 different weight matrices.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../shapeformer_amd/csrc dgemm_lds_chain.hip -o /tmp/dlc && /tmp/dlc 64
#include "../../shapeformer_amd/csrc/gpt.hip"
#include "../../shapeformer_amd/csrc/capi.hip"
#include "dg_ablation.h"
#include "dgemm_lds.hip"
#include <cmath>
#include <cstdio>
#include <vector>
template <typename F>
float time_graph(hipStream_t st, int chain, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < chain; ++i) launch(i);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (reps * chain);
}
static unsigned rs = 12345u;
static float frand() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 64;
  hipStream_t st; (void)hipStreamCreate(&st);
  const size_t MB = 1 << 20;
  const int NBUF = 24;
  std::vector<float*> bufs(NBUF);
  std::vector<float> h(4 * MB);
  for (auto& v : h) v = frand() * 0.05f;
  for (auto& b : bufs) { (void)hipMalloc(&b, 17 * MB); (void)hipMemcpy(b, h.data(), 16 * MB, hipMemcpyHostToDevice); (void)hipMemset((char*)b + 16 * MB, 0, MB); }
  float *x, *out, *out2, *res, *c1, *c2, *slab; int* cnt;
  (void)hipMalloc(&x, 16 * MB); for (auto& v : h) v = frand(); (void)hipMemcpy(x, h.data(), 16 * MB, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 16 * MB); (void)hipMalloc(&out2, 16 * MB); (void)hipMalloc(&res, 16 * MB);
  for (auto& v : h) v = frand(); (void)hipMemcpy(res, h.data(), 16 * MB, hipMemcpyHostToDevice);
  (void)hipMalloc(&c1, MB); (void)hipMalloc(&c2, MB);
  for (size_t i = 0; i < MB / 4; ++i) h[i] = frand(); (void)hipMemcpy(c1, h.data(), MB, hipMemcpyHostToDevice);
  for (size_t i = 0; i < MB / 4; ++i) h[i] = frand(); (void)hipMemcpy(c2, h.data(), MB, hipMemcpyHostToDevice);
  (void)hipMalloc(&slab, 64 * MB); (void)hipMalloc(&cnt, MB); (void)hipMemset(cnt, 0, MB);
  struct C { const char* nm; int N, K, ln, act, resid, packed, S0; } cs[] = {
      {"fc1 ln gelu", 4096, 1024, 1, 1, 0, 1, 1}, {"qkv ln", 3072, 1024, 1, 0, 0, 1, 1}, {"fc2 resid", 1024, 4096, 0, 0, 1, 1, 4},
      {"proj resid", 1024, 1024, 0, 0, 1, 1, 4}, {"head ln", 4097, 1024, 1, 0, 0, 0, 1}};
#ifdef DL_SHORT
  struct V { int NB, KP, S; } vs[] = {{4, 4, 4}, {2, 4, 2}, {2, 4, 8}, {2, 8, 8}, {3, 4, 4}, {1, 8, 4}};
  struct V2 { int NB, KP, S; } vs_unused[] = {{4, 4, 4}, {4, 2, 4}, {4, 2, 8}, {2, 4, 2}, {2, 4, 4}, {2, 8, 2}, {3, 4, 4}, {3, 4, 2}, {1, 8, 1}, {1, 8, 2}, {1, 8, 4}, {1, 8, 8},
                                       {4, 4, 16}, {4, 2, 16}, {2, 8, 8}, {2, 4, 8}, {4, 4, 8}, {2, 4, 16}, {2, 8, 4}};
#else
  struct V { int NB, KP, S; } vs[] = {{4, 4, 4}, {4, 2, 4}, {4, 2, 8}, {2, 4, 2}, {2, 4, 4}, {2, 8, 2}, {3, 4, 4}, {3, 4, 2}, {1, 8, 1}, {1, 8, 2}, {1, 8, 4}, {1, 8, 8},
                                       {4, 4, 16}, {4, 2, 16}, {2, 8, 8}, {2, 4, 8}, {4, 4, 8}, {2, 4, 16}, {2, 8, 4}};
#endif
  std::vector<float> ha(16 * MB / 4), hb(16 * MB / 4);
  for (auto c : cs) {
    const int ldo = c.packed ? c.N : 4128;
    auto base = [&](int i, float* o) {
      return dg_call(x, bufs[i % NBUF], c.ln ? c1 : nullptr, c2, c.resid ? res : nullptr, o, M, c.N, c.K, ldo, c.ln, c.act, c.packed, c.S0, slab, cnt, st);
    };
    (void)hipMemsetAsync(out, 0, 16 * MB, st);
    int rc = base(0, out);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(ha.data(), out, 16 * MB, hipMemcpyDeviceToHost);
    float t = time_graph(st, 48, 10, [&](int i) { base(i, out); });
    printf("%-12s M=%d  base S=%d             : %6.2f us (%.2f TB/s weights, %5.1f TFLOP/s) rc=%d\n", c.nm, M, c.S0, t, (double)c.N * c.K * 4 / t / 1e6,
           2.0 * M * c.N * c.K / t / 1e6, rc);
    for (auto v : vs) {
      if (c.K % (v.S * 16 * v.KP)) continue;
      const int SW = c.K / v.S / 16 / v.KP;
      if (SW != 2 && SW != 4 && SW != 8) continue;
      const int wgs = ((c.N + 15) / 16 + v.NB - 1) / v.NB * v.S;
      if (wgs < 128 || wgs > 1100) continue;
      auto nw = [&](int i, float* o) {
        return sfmi_decode_gemm_lds_f32(x, bufs[i % NBUF], c.ln ? c1 : nullptr, c2, c.resid ? res : nullptr, o, M, c.N, c.K, ldo, c.ln, c.act, c.packed, v.S, v.NB,
                                        v.KP, slab, cnt, st);
      };
      (void)hipMemsetAsync(out2, 0, 16 * MB, st);
      rc = nw(0, out2);
      if (rc) { continue; }
      hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) { printf("  NB=%d KP=%d S=%d: runtime error %d\n", v.NB, v.KP, v.S, (int)e); return 1; }
      (void)hipMemcpy(hb.data(), out2, 16 * MB, hipMemcpyDeviceToHost);
      double md = 0, mx = 0;
      const size_t n = c.packed ? (size_t)((M + 15) / 16 * 16) * c.N : (size_t)M * ldo;
      for (size_t i = 0; i < n; ++i) { md = fmax(md, fabs((double)ha[i] - hb[i])); mx = fmax(mx, fabs((double)ha[i])); }
      t = time_graph(st, 48, 10, [&](int i) { nw(i, out2); });
      printf("%-12s M=%d  lds NB=%d KP=%d SW=%d S=%2d (%4d WGs): %6.2f us (%.2f TB/s, %5.1f TFLOP/s)  max|diff| %.2e (scale %.1f)\n", c.nm, M, v.NB, v.KP, SW, v.S, wgs, t,
             (double)c.N * c.K * 4 / t / 1e6, 2.0 * M * c.N * c.K / t / 1e6, md, mx);
    }
  }
  return 0;
}
