// Micro-benchmark: what does a grid-wide barrier cost inside ONE persistent kernel on MI355X (8 XCDs, L2 per XCD)?
// The question behind it: would a persistent per-layer decode kernel (4 dependent GEMM phases, weights prefetched across the
// dependency) beat 4 graph launches at ~1.7 us of launch boundary each?  Variants: bare barrier (one agent-scope atomic + spin),
// barrier + data hand-off (every thread writes 16 B, agent-scope release / acquire, then reads another workgroup's data).
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void grid_sync(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// XCD-hierarchical barrier (round 4; the form MI355X_MICROARCH.md prices at 4.1 / 5.9 us for 256 / 512 workgroups): workgroups arrive
// on their OWN XCD's counter (32 arrivers per line instead of 256), the last arriver of an XCD arrives on the top counter, waits
// for all 8 XCDs and then releases its XCD through a per-XCD generation word; everyone polls with relaxed agent-scope loads +
// s_sleep and takes ONE acquire fence after the wait; lane 0 takes a release fence before it arrives.  Workgroup -> XCD = block % 8.
struct XcdBar { unsigned cnt[8][32]; unsigned gen[8][32]; unsigned top[32]; };
__device__ __forceinline__ void xcd_sync(XcdBar* b, unsigned g /*1, 2, ...*/, unsigned per_xcd) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned old = __hip_atomic_fetch_add(&b->cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == g * per_xcd - 1u) {
      __hip_atomic_fetch_add(&b->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // every spin is bounded (50 ms): a bug must not hang the box
      while (__hip_atomic_load(&b->top[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g * 8u &&
             __builtin_amdgcn_s_memrealtime() - t0 < 5000000ull) __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(&b->gen[x][0], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(&b->gen[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g &&
             __builtin_amdgcn_s_memrealtime() - t0 < 5000000ull) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}
template <int MODE>
__global__ __launch_bounds__(256) void k_bar_xcd(XcdBar* bar, f32x4* data, const f32x4* w, int nbar, float* out) {
  const unsigned G = gridDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nbar; ++b) {
    if (MODE >= 1) {
      f32x4 v = {(float)b, (float)blockIdx.x, (float)threadIdx.x, 1.f};
      data[((size_t)(b & 1) * G + blockIdx.x) * 256 + threadIdx.x] = v;
    }
    if (MODE == 2) {
      const f32x4* p = w + ((size_t)(b % 24) * G + blockIdx.x) * 4096 + threadIdx.x;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += p[u * 256];
    }
    xcd_sync(bar, (unsigned)(b + 1), G / 8);
    if (MODE >= 1) {
      // the consumer's loads must not be served from a stale L2 line of ANOTHER XCD's data: agent-scope (sc1) loads
      const float* src = reinterpret_cast<const float*>(&data[((size_t)(b & 1) * G + (blockIdx.x + 37) % G) * 256 + threadIdx.x]);
      f32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (r[0] != (float)b) acc[3] += 1e9f;   // stale data would show up in out[]
      acc += r;
    }
  }
  if (MODE >= 1 || acc[0] == 123.f) out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>   // 0 bare, 1 hand-off of 16 B per thread, 2 hand-off + 64 KB of streamed reads per workgroup between barriers
__global__ __launch_bounds__(256) void k_bar(unsigned* ctr, f32x4* data, const f32x4* w, int nbar, float* out) {
  const unsigned G = gridDim.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nbar; ++b) {
    if (MODE >= 1) {
      f32x4 v = {(float)b, (float)blockIdx.x, (float)threadIdx.x, 1.f};
      data[((size_t)(b & 1) * G + blockIdx.x) * 256 + threadIdx.x] = v;
      __threadfence();
    }
    if (MODE == 2) {
      const f32x4* p = w + ((size_t)(b % 24) * G + blockIdx.x) * 4096 + threadIdx.x;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += p[u * 256];
    }
    grid_sync(ctr, (unsigned)(b + 1) * G);
    if (MODE >= 1) {
      const f32x4 r = data[((size_t)(b & 1) * G + (blockIdx.x + 37) % G) * 256 + threadIdx.x];
      if (r[0] != (float)b) acc[3] += 1e9f;   // stale data would show up in out[]
      acc += r;
    }
  }
  if (MODE >= 1 || acc[0] == 123.f) out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main(int argc, char** argv) {
  const int nbar = 2000;
  unsigned* ctr; f32x4 *data, *w; float* out;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&data, 2 * 1024 * 256 * 16)); CK(hipMalloc(&out, 1024 * 256 * 4));
  CK(hipMalloc(&w, (size_t)24 * 1024 * 4096 * 16)); CK(hipMemset(w, 0, (size_t)24 * 1024 * 4096 * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int G : {64, 256, 512}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4));
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(G), dim3(256), 0, 0, ctr, data, w, nbar, out);
        if (mode == 1) hipLaunchKernelGGL(k_bar<1>, dim3(G), dim3(256), 0, 0, ctr, data, w, nbar, out);
        if (mode == 2) hipLaunchKernelGGL(k_bar<2>, dim3(G), dim3(256), 0, 0, ctr, data, w, nbar, out);
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      float h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      {   // the same with the XCD-hierarchical barrier
        XcdBar* xb; CK(hipMalloc(&xb, sizeof(XcdBar)));
        float bx = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(xb, 0, sizeof(XcdBar)));
          hipEventRecord(e0, 0);
          if (mode == 0) hipLaunchKernelGGL(k_bar_xcd<0>, dim3(G), dim3(256), 0, 0, xb, data, w, nbar, out);
          if (mode == 1) hipLaunchKernelGGL(k_bar_xcd<1>, dim3(G), dim3(256), 0, 0, xb, data, w, nbar, out);
          if (mode == 2) hipLaunchKernelGGL(k_bar_xcd<2>, dim3(G), dim3(256), 0, 0, xb, data, w, nbar, out);
          hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < bx) bx = ms;
        }
        float hx[4]; CK(hipMemcpy(hx, out, 16, hipMemcpyDeviceToHost));
        printf("G=%4d mode %d XCD-hierarchical barrier: %.2f us per barrier%s\n", G, mode, bx * 1e3f / nbar, (mode && hx[0] > 1e8f) ? "  STALE DATA" : "");
        (void)hipFree(xb);
      }
      printf("G=%4d mode %d (%s): %.2f us per barrier%s\n", G, mode, mode == 0 ? "bare" : mode == 1 ? "16 B/thread hand-off" : "hand-off + 64 KB stream per WG",
             best * 1e3f / nbar, (mode && h[0] > 1e8f) ? "  STALE DATA" : "");
    }
  }
  return 0;
}
