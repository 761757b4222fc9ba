// Throughput of the decode GEMM phase under chain interleave: NS streams, each a hipGraph of 24 "layers" (qkv, proj, fc1, fc2 at M rows)
// replayed back to back - the r1 kernel (csrc/gpt.hip) against LDS-staged shapes (dgemm_lds.hip).  us per chain-layer.
#include "../../shapeformer_amd/csrc/gpt.hip"
#include "../../shapeformer_amd/csrc/capi.hip"
#include "dg_ablation.h"
#include "dgemm_lds.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 64, NS = argc > 2 ? atoi(argv[2]) : 3;
  const size_t MB = 1 << 20;
  const int L = 24;
  std::vector<float*> wq(L), wp(L), w1(L), w2(L);
  for (int l = 0; l < L; ++l) {
    (void)hipMalloc(&wq[l], 12 * MB); (void)hipMalloc(&wp[l], 4 * MB); (void)hipMalloc(&w1[l], 16 * MB); (void)hipMalloc(&w2[l], 16 * MB);
    (void)hipMemset(wq[l], 0, 12 * MB); (void)hipMemset(wp[l], 0, 4 * MB); (void)hipMemset(w1[l], 0, 16 * MB); (void)hipMemset(w2[l], 0, 16 * MB);
  }
  float *c1, *c2; (void)hipMalloc(&c1, MB); (void)hipMalloc(&c2, MB); (void)hipMemset(c1, 0, MB); (void)hipMemset(c2, 0, MB);
  struct Cfg { const char* nm; int lds; int q[3], p[3], f1[3], f2[3]; } cfgs[] = {
      {"r1 dgemm_kernel (qkv S1, proj S4, fc1 S1, fc2 S4)", 0, {0, 0, 1}, {0, 0, 4}, {0, 0, 1}, {0, 0, 4}},
      {"lds: qkv 3,4,S4  proj r1  fc1 2,4,S2  fc2 2,4,S8", 1, {3, 4, 4}, {0, 0, 4}, {2, 4, 2}, {2, 4, 8}},
      {"lds: qkv 3,4,S4  proj r1  fc1 4,4,S4  fc2 2,8,S8", 1, {3, 4, 4}, {0, 0, 4}, {4, 4, 4}, {2, 8, 8}},
      {"lds: qkv r1      proj r1  fc1 r1      fc2 2,4,S8", 1, {0, 0, 1}, {0, 0, 4}, {0, 0, 1}, {2, 4, 8}},
      {"lds: qkv 3,4,S4  proj r1  fc1 2,4,S2  fc2 r1", 1, {3, 4, 4}, {0, 0, 4}, {2, 4, 2}, {0, 0, 4}},
  };
  for (auto& cf : cfgs) {
#ifdef DGS_R1_ONLY
    if (cf.lds) continue;
#endif
    std::vector<hipStream_t> st(NS);
    std::vector<hipGraphExec_t> ge(NS);
    std::vector<float*> x(NS), qkv(NS), y(NS), r(NS), h(NS), slab(NS);
    std::vector<int*> cnt(NS);
    for (int s = 0; s < NS; ++s) {
      (void)hipStreamCreate(&st[s]);
      (void)hipMalloc(&x[s], MB); (void)hipMalloc(&qkv[s], 2 * MB); (void)hipMalloc(&y[s], MB); (void)hipMalloc(&r[s], MB); (void)hipMalloc(&h[s], 2 * MB);
      (void)hipMemset(r[s], 0, MB); (void)hipMemset(y[s], 0, MB); (void)hipMemset(h[s], 0, 2 * MB);
      (void)hipMalloc(&slab[s], 32 * MB); (void)hipMalloc(&cnt[s], MB); (void)hipMemset(cnt[s], 0, MB);
      auto G = [&](const int* v, const float* X, const float* W, const float* C1, const float* Rs, float* O, int N, int K, int ln, int act) {
        if (v[0] == 0) dg_call(X, W, C1, c2, Rs, O, M, N, K, N, ln, act, 1, v[2], slab[s], cnt[s], st[s]);
        else sfmi_decode_gemm_lds_f32(X, W, C1, c2, Rs, O, M, N, K, N, ln, act, 1, v[2], v[0], v[1], slab[s], cnt[s], st[s]);
      };
      hipGraph_t g;
      (void)hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal);
      for (int l = 0; l < L; ++l) {
        G(cf.q, r[s], wq[l], c1, nullptr, qkv[s], 3072, 1024, 1, 0);
        G(cf.p, y[s], wp[l], nullptr, r[s], r[s], 1024, 1024, 0, 0);
        G(cf.f1, r[s], w1[l], c1, nullptr, h[s], 4096, 1024, 1, 1);
        G(cf.f2, h[s], w2[l], nullptr, r[s], r[s], 1024, 4096, 0, 0);
      }
      (void)hipStreamEndCapture(st[s], &g);
      (void)hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0);
    }
    const int reps = 20;
    for (int s = 0; s < NS; ++s) (void)hipGraphLaunch(ge[s], st[s]);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st[0]);
    for (int rp = 0; rp < reps; ++rp)
      for (int s = 0; s < NS; ++s) (void)hipGraphLaunch(ge[s], st[s]);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e1, st[0]); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("M=%d x %d chains  %-52s: %.2f us per chain-layer (%.1f TFLOP/s)\n", M, NS, cf.nm, ms * 1e3 / (reps * NS * L),
           2.0 * M * 12.58e6 * reps * NS * L / (ms * 1e-3) / 1e12);
    for (int s = 0; s < NS; ++s) { (void)hipStreamDestroy(st[s]); (void)hipFree(x[s]); (void)hipFree(qkv[s]); (void)hipFree(y[s]); (void)hipFree(r[s]); (void)hipFree(h[s]); (void)hipFree(slab[s]); (void)hipFree(cnt[s]); }
  }
  return 0;
}
