"""Prefill / training GEMM shapes: sfmi_gemm_f32 (conv3d_igemm tile kernel) against the library sgemm torch dispatches to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L
from bench import ev_time
dev = torch.device("cuda:0")
lib = L.lib()
for M, N, K in ((10048, 3072, 1024), (10048, 1024, 1024), (10048, 4096, 1024), (10048, 1024, 4096), (30144, 4096, 1024), (4000, 1024, 4096), (1024, 4096, 10048)):
    x, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    f1 = lambda: L.check(lib.sfmi_gemm_f32(L.ptr(x), L.ptr(W), L.ptr(b), None, L.ptr(y), M, N, K, 0, 0, 0, L.stream_ptr()), "gemm")
    f2 = lambda: torch.addmm(b, x, W.t(), out=y)
    f3 = lambda: L.check(lib.sfmi_gemm_blas_f32(L.ptr(x), L.ptr(W), L.ptr(b), None, L.ptr(y), M, N, K, 0, L.stream_ptr()), "gemm_blas")
    f1(); f2(); f3(); torch.cuda.synchronize()
    t1, t2, t3 = ev_time(f1, 10), ev_time(f2, 10), ev_time(f3, 10)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: sfmi {t1*1e3:8.1f} us {fl/t1/1e9:6.1f} TF | torch.addmm {t2*1e3:8.1f} us {fl/t2/1e9:6.1f} TF | sfmi_gemm_blas {t3*1e3:8.1f} us {fl/t3/1e9:6.1f} TF")
