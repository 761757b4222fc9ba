// Does a weight matrix that was just streamed (by a prefetch kernel) come back faster than from HBM?  Reads a 48 MB
// buffer (larger than the 8 x 4 MB L2s, smaller than the 256 MB Infinity Cache) in three regimes inside a hipGraph chain:
//   cold    : rotate over 12 x 48 MB buffers (576 MB > Infinity Cache) -> every read comes from HBM
//   warm    : the same buffer every time                              -> Infinity Cache hits if reads allocate there
//   prefetch: kernel A (64 workgroups, low priority stand-in) reads buffer i+1 while kernel B reads buffer i from a second stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int UN, bool NT>
__global__ void k_stream(const f32x4* __restrict__ w, float* __restrict__ o, int loads_per_wave) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const f32x4* p = w + wave * loads_per_wave * 64 + lane;
  f32x4 acc = {0, 0, 0, 0};
  for (int s = 0; s < loads_per_wave; s += UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(p + (s + u) * 64) : p[(s + u) * 64];
#pragma unroll
    for (int u = 0; u < UN; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) o[wave] = acc[0];
}
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
int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  const size_t MB = 1 << 20, SZ = 48 * MB;
  const int NBUF = 12;
  std::vector<f32x4*> bufs(NBUF);
  for (auto& b : bufs) { (void)hipMalloc(&b, SZ); (void)hipMemset(b, 0, SZ); }
  float* o; (void)hipMalloc(&o, 64 * MB);
  const int wgs = 384, threads = 512;   // 3072 waves x 16 KiB = 48 MB, 16 loads per wave (multiple of UN = 8)
  const long long waves = (long long)wgs * threads / 64;
  const int lpw = (int)(SZ / 1024 / waves);
  for (int nt = 0; nt < 2; ++nt) {
    auto launch = [&](f32x4* b) {
      if (nt) hipLaunchKernelGGL((k_stream<8, true>), dim3(wgs), dim3(threads), 0, st, b, o, lpw);
      else hipLaunchKernelGGL((k_stream<8, false>), dim3(wgs), dim3(threads), 0, st, b, o, lpw);
    };
    float cold = time_graph(st, 48, 10, [&](int i) { launch(bufs[i % NBUF]); });
    float warm = time_graph(st, 48, 10, [&](int i) { launch(bufs[0]); });
    float pair = time_graph(st, 48, 10, [&](int i) { launch(bufs[(i / 2) % NBUF]); });   // every buffer read twice in a row
    printf("%s loads, 48 MB: cold %.1f us (%.2f TB/s) | same buffer %.1f us (%.2f TB/s) | read-twice avg %.1f us\n", nt ? "nontemporal" : "plain",
           cold, SZ / cold / 1e6, warm, SZ / warm / 1e6, pair);
  }
  return 0;
}
