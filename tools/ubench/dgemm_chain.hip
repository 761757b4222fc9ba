// Apples-to-apples: the production decode GEMM (csrc/gpt.hip) in the same C++ hipGraph chain as stream_chain.hip
#include "../../shapeformer_amd/csrc/gpt.hip"
#include "../../shapeformer_amd/csrc/capi.hip"
#include "dg_ablation.h"
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
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16;
  hipStream_t st; (void)hipStreamCreate(&st);
  const size_t MB = 1 << 20;
  const int NBUF = 24;
  std::vector<float*> bufs(NBUF);
  for (auto& b : bufs) { (void)hipMalloc(&b, 16 * MB); (void)hipMemset(b, 0, 16 * MB); }
  float *x, *out, *c1, *c2, *slab; int* cnt;
  (void)hipMalloc(&x, 16 * MB); (void)hipMemset(x, 0, 16 * MB);
  (void)hipMalloc(&out, 16 * MB); (void)hipMalloc(&c1, 1 * MB); (void)hipMemset(c1, 0, MB); (void)hipMalloc(&c2, 1 * MB); (void)hipMemset(c2, 0, MB);
  (void)hipMalloc(&slab, 64 * MB); (void)hipMalloc(&cnt, MB); (void)hipMemset(cnt, 0, MB);
  struct C { const char* nm; int N, K, ln, act, S, resid; } cs[] = {
      {"fc1 ln gelu", 4096, 1024, 1, 1, 1, 0}, {"fc1 plain", 4096, 1024, 0, 0, 1, 0}, {"qkv ln", 3072, 1024, 1, 0, 1, 0},
      {"fc2 S1", 1024, 4096, 0, 0, 1, 1}, {"fc2 S2", 1024, 4096, 0, 0, 2, 1}, {"fc2 S4", 1024, 4096, 0, 0, 4, 1}, {"fc2 S8", 1024, 4096, 0, 0, 8, 1}, {"proj S1", 1024, 1024, 0, 0, 1, 1}, {"proj S2", 1024, 1024, 0, 0, 2, 1}, {"proj S4", 1024, 1024, 0, 0, 4, 1}, {"head", 4097, 1024, 1, 0, 1, 0}};
  for (auto c : cs) {
    float t = time_graph(st, 48, 10, [&](int i) {
      dg_call(x, bufs[i % NBUF], c.ln ? c1 : nullptr, c2, c.resid ? out : nullptr, out, M, c.N, c.K, c.N, c.ln, c.act, 1, c.S, slab, cnt, st);
    });
    printf("%-12s M=%d: %.2f us  (%.2f TB/s weights, %.1f TFLOP/s)\n", c.nm, M, t, (double)c.N * c.K * 4 / t / 1e6, 2.0 * M * c.N * c.K / t / 1e6);
  }
  return 0;
}
