"""Weight-gradient shaped products dW (N,K) = dY^T (N,M) X (M,K): rocBLAS TN form vs explicit transposes + the tile kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L
from bench import ev_time
dev = torch.device("cuda:0")
lib = L.lib()
for M, N, K in ((3992, 1024, 1024), (3992, 3072, 1024), (3992, 4096, 1024), (3992, 1024, 4096), (499, 4096, 1024), (499, 1024, 4096)):
    dY, X = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    out = torch.empty(N, K, device=dev)
    Mp = (M + 15) // 16 * 16
    dYT, XT = torch.zeros(N, Mp, device=dev), torch.zeros(K, Mp, device=dev)
    def tile():
        L.check(lib.sfmi_transpose_f32(L.ptr(dY), L.ptr(dYT), M, N, N, Mp, L.stream_ptr()), "t")
        L.check(lib.sfmi_transpose_f32(L.ptr(X), L.ptr(XT), M, K, K, Mp, L.stream_ptr()), "t")
        L.check(lib.sfmi_gemm_f32(L.ptr(dYT), L.ptr(XT), None, None, L.ptr(out), N, K, Mp, 0, 0, 0, L.stream_ptr()), "g")
    blas = lambda: L.check(lib.sfmi_sgemm_f32(1, 0, N, K, M, 1.0, L.ptr(dY), N, L.ptr(X), K, 0.0, L.ptr(out), K, L.stream_ptr()), "s")
    th = lambda: torch.mm(dY.t(), X, out=out)
    dxb = lambda: L.check(lib.sfmi_sgemm_f32(0, 0, M, K, N, 1.0, L.ptr(dY), N, L.ptr(torch.empty(N, K, device=dev)), K, 0.0, L.ptr(X), K, L.stream_ptr()), "dx")
    for f in (tile, blas, th): f()
    torch.cuda.synchronize()
    t1, t2, t3 = ev_time(tile, 10), ev_time(blas, 10), ev_time(th, 10)
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d}: transposes+tile {t1*1e3:7.1f} us {fl/t1/1e9:6.1f} TF | rocblas TN {t2*1e3:7.1f} us {fl/t2/1e9:6.1f} TF | torch.mm {t3*1e3:7.1f} us {fl/t3/1e9:6.1f} TF")
