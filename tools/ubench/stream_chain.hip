// Micro-benchmark: what does a chain of short weight-streaming kernels cost on MI355X inside a hipGraph?
// Variants: empty kernel; pure streaming (sum) with different workgroup counts / waves / loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(float* o) { if (threadIdx.x == 9999) o[0] = 1.f; }

// each wave streams `per_wave_kb` KiB contiguous (1 KiB per load instruction), UN loads in flight
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

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
float time_graph(hipStream_t st, int chain, int reps, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < chain; ++i) launch(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1e3f / (reps * chain);  // us per kernel
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t MB = 1 << 20;
  const int NBUF = 24;               // rotate over 24 x 16 MB buffers (> 256 MB Infinity Cache)
  std::vector<f32x4*> bufs(NBUF);
  for (auto& b : bufs) { CK(hipMalloc(&b, 16 * MB)); CK(hipMemset(b, 0, 16 * MB)); }
  float* o; CK(hipMalloc(&o, 64 * MB));
  printf("empty 256x512: %.2f us/kernel\n", time_graph(st, 48, 20, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, o); }));
  printf("empty 64x64  : %.2f us/kernel\n", time_graph(st, 48, 20, [&](int) { hipLaunchKernelGGL(k_empty, dim3(64), dim3(64), 0, st, o); }));
  struct Cfg { int mb, wgs, threads; };
  Cfg cfgs[] = {{16, 256, 512}, {16, 512, 256}, {16, 1024, 256}, {16, 256, 1024}, {16, 128, 1024}, {16, 2048, 64}, {4, 64, 512}, {4, 256, 128}, {4, 256, 256}, {4, 512, 64}, {4, 1024, 64}, {12, 192, 512}, {12, 768, 128}};
  for (auto c : cfgs) {
    const long long waves = (long long)c.wgs * c.threads / 64;
    const int lpw = (int)((long long)c.mb * MB / 1024 / waves);
    float t1 = lpw % 8 ? -1.f : time_graph(st, 48, 10, [&](int i) { hipLaunchKernelGGL((k_stream<8, false>), dim3(c.wgs), dim3(c.threads), 0, st, bufs[i % NBUF], o, lpw); });
    float t2 = lpw % 8 ? -1.f : time_graph(st, 48, 10, [&](int i) { hipLaunchKernelGGL((k_stream<8, true>), dim3(c.wgs), dim3(c.threads), 0, st, bufs[i % NBUF], o, lpw); });
    float t3 = lpw % 16 == 0 ? time_graph(st, 48, 10, [&](int i) { hipLaunchKernelGGL((k_stream<16, false>), dim3(c.wgs), dim3(c.threads), 0, st, bufs[i % NBUF], o, lpw); }) : -1.f;
    float t4 = lpw % 4 ? -1.f : time_graph(st, 48, 10, [&](int i) { hipLaunchKernelGGL((k_stream<4, false>), dim3(c.wgs), dim3(c.threads), 0, st, bufs[i % NBUF], o, lpw); });
    printf("%2d MB  %4d WGs x %4d thr (%3d loads/wave): UN8 %.2f us (%.2f TB/s) | UN8 nt %.2f | UN16 %.2f | UN4 %.2f\n", c.mb, c.wgs, c.threads, lpw,
           t1, c.mb * MB / t1 / 1e6, t2, t3, t4);
  }
  return 0;
}
