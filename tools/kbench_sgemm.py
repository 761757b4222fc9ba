"""csrc/sgemm.hip against the r1 tile kernel (conv3d_igemm as a GEMM) and the library sgemm, prefill / training shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L
from bench import ev_time
dev = torch.device("cuda:0")
lib = L.lib()
print("forward y = x W^T (+bias):")
for M, N, K in ((30144, 3072, 1024), (30144, 1024, 1024), (30144, 4096, 1024), (30144, 1024, 4096), (10048, 4096, 1024), (3992, 4096, 1024), (3992, 1024, 4096), (3992, 4128, 1024)):
    x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    f0 = lambda: L.check(lib.sfmi_sgemm_mfma_f32(0, 1, M, N, K, L.ptr(x), K, L.ptr(W), K, L.ptr(y), N, 0, L.ptr(b), 0, None, None, 0, 0.0, 0, L.stream_ptr()), "sgemm_mfma")
    f1 = lambda: L.check(lib.sfmi_gemm_f32(L.ptr(x), L.ptr(W), L.ptr(b), None, L.ptr(y), M, N, K, 0, 0, 0, L.stream_ptr()), "gemm")
    f3 = lambda: L.check(lib.sfmi_gemm_blas_f32(L.ptr(x), L.ptr(W), L.ptr(b), None, L.ptr(y), M, N, K, 0, L.stream_ptr()), "gemm_blas")
    fs = [f0, f1, f3] if N % 32 == 0 else [f0, f0, f3]
    for f in fs: f()
    torch.cuda.synchronize()
    t = [ev_time(f, 10) for f in fs]
    fl = 2.0 * M * N * K
    print(f"  M={M:6d} N={N:5d} K={K:5d}: sgemm_mfma {t[0]*1e3:8.1f} us {fl/t[0]/1e9:6.1f} TF | r1 tile {t[1]*1e3:8.1f} us {fl/t[1]/1e9:6.1f} TF | rocBLAS+epilogue {t[2]*1e3:8.1f} us {fl/t[2]/1e9:6.1f} TF")
print("dX = dY W  (A (M,N) K-contiguous, B = W (N,K) stored (k,n)):")
for M, N, K in ((3992, 3072, 1024), (3992, 4096, 1024), (3992, 1024, 4096), (3992, 1024, 1024)):
    dY, W = torch.randn(M, N, device=dev), torch.randn(N, K, device=dev)
    dx = torch.empty(M, K, device=dev)
    f0 = lambda: L.check(lib.sfmi_sgemm_mfma_f32(0, 0, M, K, N, L.ptr(dY), N, L.ptr(W), K, L.ptr(dx), K, 0, None, 0, None, None, 0, 0.0, 0, L.stream_ptr()), "dx")
    f3 = lambda: L.check(lib.sfmi_sgemm_f32(0, 0, M, K, N, 1.0, L.ptr(dY), N, L.ptr(W), K, 0.0, L.ptr(dx), K, L.stream_ptr()), "dx blas")
    for f in (f0, f3): f()
    torch.cuda.synchronize()
    t = [ev_time(f, 10) for f in (f0, f3)]
    fl = 2.0 * M * N * K
    print(f"  M={M:5d} N={N:5d} K={K:5d}: sgemm_mfma {t[0]*1e3:8.1f} us {fl/t[0]/1e9:6.1f} TF | rocBLAS NN {t[1]*1e3:8.1f} us {fl/t[1]/1e9:6.1f} TF")
print("dW = dY^T X  (A = dY stored (k=m, n), B = X stored (k=m, kk)):")
for M, N, K in ((3992, 1024, 1024), (3992, 3072, 1024), (3992, 4096, 1024), (3992, 1024, 4096), (499, 4096, 1024)):
    dY, X = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    out = torch.empty(N, K, device=dev)
    ws = torch.empty(lib.sfmi_sgemm_mfma_splits(N, K, M) * N * K, device=dev)
    f0 = lambda: L.check(lib.sfmi_sgemm_mfma_f32(1, 0, N, K, M, L.ptr(dY), N, L.ptr(X), K, L.ptr(out), K, 0, None, 0, None, L.ptr(ws), ws.numel(), 0.0, 0, L.stream_ptr()), "dw")
    f3 = lambda: L.check(lib.sfmi_sgemm_f32(1, 0, N, K, M, 1.0, L.ptr(dY), N, L.ptr(X), K, 0.0, L.ptr(out), K, L.stream_ptr()), "dw blas")
    for f in (f0, f3): f()
    torch.cuda.synchronize()
    t = [ev_time(f, 10) for f in (f0, f3)]
    fl = 2.0 * M * N * K
    print(f"  M={M:5d} N={N:5d} K={K:5d}: sgemm_mfma {t[0]*1e3:8.1f} us {fl/t[0]/1e9:6.1f} TF | rocBLAS TN {t[1]*1e3:8.1f} us {fl/t[1]/1e9:6.1f} TF")
