// Library sgemm for the PLAIN large GEMMs of the path (prefill of the condition prefix and the training step of the
// transformer: M = thousands of rows, mingpt.py:46-111 Linear layers and their gradients).  The fused / skinny hot ops
// (decode GEMMs, convolutions, SDF query, weight-gradient of the autoencoder) stay hand-written; for a plain
// M x N x K product rocBLAS's tuned gfx950 f32 kernels reach 120-146 TFLOP/s where the 128x128 tile kernel of
// conv3d.hip reaches 80-107 (tools/kbench_gemm.py), and they take transposed operands directly, which removes the explicit
// transposes from the weight-gradient GEMMs.
//
// rocBLAS is bound lazily with dlopen("librocblas.so.5") - the copy PyTorch-ROCm already mapped when the caller is a
// torch process (same SONAME), the system one otherwise - so libsfmi.so carries no link-time dependency; without it
// sfmi_sgemm_f32 returns SFMI_ENOBLAS and sfmi_gemm_f32 keeps using the tile kernel.  State: an immutable
// function-pointer table (initialised once, thread-safe) and one rocblas_handle per host thread.
#include "sfmi_common.h"
#include <dlfcn.h>

#define SFMI_ENOBLAS (-3)

namespace {

typedef struct _rocblas_handle* rb_handle;
typedef int (*rb_create_t)(rb_handle*);
typedef int (*rb_set_stream_t)(rb_handle, hipStream_t);
typedef int (*rb_sgemm_t)(rb_handle, int, int, int, int, int, const float*, const float*, int, const float*, int, const float*, float*, int);

struct RB { rb_create_t create; rb_set_stream_t set_stream; rb_sgemm_t sgemm; };

RB load_rb() {
  RB r = {nullptr, nullptr, nullptr};
  void* lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return r;
  r.create = (rb_create_t)dlsym(lib, "rocblas_create_handle");
  r.set_stream = (rb_set_stream_t)dlsym(lib, "rocblas_set_stream");
  r.sgemm = (rb_sgemm_t)dlsym(lib, "rocblas_sgemm");
  if (!r.create || !r.set_stream || !r.sgemm) r = RB{nullptr, nullptr, nullptr};
  return r;
}

const RB& rb() { static const RB r = load_rb(); return r; }

rb_handle thread_handle() {
  static thread_local rb_handle h = nullptr;
  if (!h && rb().create && rb().create(&h) != 0) h = nullptr;
  return h;
}

// y = act(y + bias) + resid on (M,N) rows
__global__ void gemm_epilogue_kernel(float* __restrict__ y, const float* __restrict__ bias, const float* __restrict__ resid, int act,
                                     int N4, long long total4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  f32x4 v = reinterpret_cast<f32x4*>(y)[i];
  if (bias) v = v + reinterpret_cast<const f32x4*>(bias)[i % N4];
  if (act == 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
  }
  if (resid) v = v + reinterpret_cast<const f32x4*>(resid)[i];
  reinterpret_cast<f32x4*>(y)[i] = v;
}

}  // namespace

extern "C" {

int sfmi_blas_available(void) { return rb().sgemm != nullptr && thread_handle() != nullptr; }

// Row-major C (M,N; ldc) = alpha * op(A) * op(B) + beta * C with op(A) (M,K), op(B) (K,N); transX != 0 means the stored
// matrix is the transpose (A stored (K,M; lda), B stored (N,K; ldb)).
int sfmi_sgemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                   float beta, float* C, int ldc, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return SFMI_EINVAL;
  rb_handle h = thread_handle();
  if (!h) return SFMI_ENOBLAS;
  if (rb().set_stream(h, (hipStream_t)stream) != 0) return SFMI_ENOBLAS;
  // row-major product == column-major product of the swapped operands: C^T = op(B)^T op(A)^T
  const int opN = 111, opT = 112;
  const int st = rb().sgemm(h, transB ? opT : opN, transA ? opT : opN, N, M, K, &alpha, B, ldb, A, lda, &beta, C, ldc);
  return st == 0 ? SFMI_OK : SFMI_ENOBLAS;
}

// y (M,N) = act(x (M,K) W (N,K)^T + bias) + resid through the library product + one fused epilogue pass.
// resid may alias y only when act == 0 (then the product accumulates into y).  Returns SFMI_ENOBLAS when rocBLAS is absent.
int sfmi_gemm_blas_f32(const float* x, const float* W, const float* bias, const float* resid, float* y, int M, int N, int K,
                       int act, void* stream) {
  if (!x || !W || !y || N % 4 || (resid == y && act != 0)) return SFMI_EINVAL;
  const bool inplace = resid == y;
  const int rc = sfmi_sgemm_f32(0, 1, M, N, K, 1.0f, x, K, W, K, inplace ? 1.0f : 0.0f, y, N, stream);
  if (rc != SFMI_OK) return rc;
  const float* r2 = inplace ? nullptr : resid;
  if (bias || act || r2) {
    const long long total4 = (long long)M * (N / 4);
    hipLaunchKernelGGL(gemm_epilogue_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, bias, r2, act,
                       N / 4, total4);
    SFMI_CHECK_LAUNCH();
  }
  return SFMI_OK;
}

}  // extern "C"
