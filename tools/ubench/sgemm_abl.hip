// csrc/sgemm.hip main-loop ablations (timing only): which part of the loop keeps the kernel at ~0.75 of the f32 MFMA peak.
#include "../../shapeformer_amd/csrc/sgemm.hip"
#include <cstdio>
int main() {
  const int M = 10048, N = 4096, K = 1024;
  float *A, *B, *C;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMemset(A, 0, (size_t)M * K * 4); (void)hipMemset(B, 0, (size_t)N * K * 4);
  hipStream_t st; (void)hipStreamCreate(&st);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) sfmi_sgemm_mfma_f32(0, 1, M, N, K, A, K, B, K, C, N, 0, nullptr, 0, nullptr, nullptr, 0, st);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < 10; ++r) sfmi_sgemm_mfma_f32(0, 1, M, N, K, A, K, B, K, C, N, 0, nullptr, 0, nullptr, nullptr, 0, st);
  (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f us  %.1f TFLOP/s\n", SG_NAME, ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
  return 0;
}
