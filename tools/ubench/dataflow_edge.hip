// Micro-benchmark (round 6): what does ONE dependency edge of BASELINE config 3's 16-row decode layer cost when it
// is kept inside a persistent launch as a tagged-granule hand-off (MI355X_MICROARCH.md "allgather" row: 8-byte {data, tag} granules,
// one sc1 store per granule, consumers sweep with sc1 loads and spin on the tags - no barrier), against the same edge as a kernel
// boundary (producer launch, consumer launch, operands re-read through L2)?
//
// The edge: an activation tensor of 16 rows x K floats (K = 1024: the residual stream / attention output, 64 KB; K = 4096: the MLP
// hidden layer, 256 KB) is produced column-sliced by all G = 256 workgroups (one per CU) and EVERY workgroup of the next GEMM needs all
// of it (a 16-column n-tile multiplies the full K).  Four of the five edges of a layer have this shape (attention -> proj, proj ->
// fc1, fc1 -> fc2, fc2 -> next qkv); the batch-1 layer of the guide hands over 1/16 of these bytes.
//
//   hipcc --offload-arch=gfx950 -O3 dataflow_edge.hip -o /tmp/dfe && /tmp/dfe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- in-launch edge: tagged granules ------------------------------------------------------------------------------------------
// buffer [2][NG] of 8-byte granules {float bits, tag}; iteration i (tag i + 1) uses half i & 1.  A workgroup publishes its NG / G
// granules, then gathers all NG of them (every lane spins on its own granules' tags), adds them up (stand-in for staging them as a
// GEMM operand) and starts the next iteration.  A producer can run at most one iteration ahead of the slowest consumer (it has to
// gather everybody's iteration i before it publishes i + 1), so two halves are enough.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void edge_granules(unsigned long long* buf, int NG, int iters, float* out, const f32x4* w, int wstream) {
  const int G = gridDim.x, per = NG / G, tid = threadIdx.x;
  float acc = 0.f;
  f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    unsigned long long* half = buf + (size_t)(it & 1) * NG;
    const unsigned tag = (unsigned)it + 1u;
    if (wstream) {      // the weight stream a real layer keeps going meanwhile: 64 KB per workgroup and edge (16.7 MB per edge, 50 MB per layer / 3)
      const f32x4* p = w + ((size_t)(it % 16) * G + blockIdx.x) * 4096 + tid;
#pragma unroll
      for (int u = 0; u < 4096 / THREADS; ++u) wacc += __builtin_nontemporal_load(p + u * THREADS);
    }
    for (int i = tid; i < per; i += THREADS) {
      const float v = (float)(it & 7) + 0.001f * (float)(blockIdx.x * per + i);
      const unsigned long long g = ((unsigned long long)tag << 32) | __float_as_uint(v);
      __hip_atomic_store(half + (size_t)blockIdx.x * per + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // one sc1 8-byte store
    }
    // gather: 8 granules in flight per lane per sweep
    for (int i0 = tid; i0 < NG; i0 += 8 * THREADS) {
      unsigned long long g[8];
      bool done = false;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (!done) {
        done = true;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * THREADS;
          g[u] = i < NG ? __hip_atomic_load(half + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) done = done && (unsigned)(g[u] >> 32) == tag;
        if (!done && __builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { acc += 1e9f; break; }      // bounded spin: a bug must not hang the box
        if (!done) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += __uint_as_float((unsigned)g[u]);
    }
    __syncthreads();      // the workgroup's operand is complete (LDS staging would end here)
  }
  out[blockIdx.x * THREADS + tid] = acc + wacc[0] + wacc[1] + wacc[2] + wacc[3];
}

// ---- the same edge as a kernel boundary: producer launch writes its slice (plain stores), consumer launch reads all of it -------------
template <int THREADS>
__global__ __launch_bounds__(THREADS) void edge_launch(const float* src, float* dst, int N, int it, float* out, const f32x4* w, int wstream) {
  const int G = gridDim.x, per = N / G, tid = threadIdx.x;
  f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
  if (wstream) {
    const f32x4* p = w + ((size_t)(it % 16) * G + blockIdx.x) * 4096 + tid;
#pragma unroll
    for (int u = 0; u < 4096 / THREADS; ++u) wacc += __builtin_nontemporal_load(p + u * THREADS);
  }
  float acc = 0.f;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  for (int i0 = tid; i0 < N / 4; i0 += 4 * THREADS) {      // the whole tensor of the previous launch, 4 x 16 B in flight per lane
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i0 + u * THREADS < N / 4 ? s4[i0 + u * THREADS] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
  }
  __syncthreads();
  for (int i = tid; i < per; i += THREADS) dst[(size_t)blockIdx.x * per + i] = acc * 1e-9f + (float)(it & 7) + 0.001f * (float)(blockIdx.x * per + i);
  if (acc == 123.456f) out[blockIdx.x * THREADS + tid] = acc + wacc[0] + wacc[1] + wacc[2] + wacc[3];
}

int main() {
  const int G = 256, iters = 2000;
  unsigned long long* buf; float *a, *b, *out; f32x4* w;
  CK(hipMalloc(&buf, 2 * 65536 * 8)); CK(hipMalloc(&a, 65536 * 4)); CK(hipMalloc(&b, 65536 * 4)); CK(hipMalloc(&out, G * 512 * 4));
  CK(hipMalloc(&w, (size_t)16 * G * 4096 * 16)); CK(hipMemset(w, 0, (size_t)16 * G * 4096 * 16));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int K : {1024, 4096}) {
    const int N = 16 * K;
    for (int ws = 0; ws < 2; ++ws) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(buf, 0, 2 * 65536 * 8));
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(edge_granules<512>, dim3(G), dim3(512), 0, 0, buf, N, iters, out, w, ws);
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      float h; CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
      float bl = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it)
          hipLaunchKernelGGL(edge_launch<512>, dim3(G), dim3(512), 0, 0, (it & 1) ? b : a, (it & 1) ? a : b, N, it, out, w, ws);
        hipEventRecord(e1, 0); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < bl) bl = ms;
      }
      printf("16 rows x K = %4d (%3d KB), %s: in-launch tagged-granule allgather %.2f us per edge%s | kernel boundary + re-read %.2f us per edge\n", K,
             N * 4 / 1024, ws ? "with a 64 KB/workgroup weight stream per edge" : "bare", best * 1e3f / iters, h > 1e8f ? "  (SPIN TIMED OUT)" : "",
             bl * 1e3f / iters);
    }
  }
  return 0;
}
