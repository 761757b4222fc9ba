"""Round-4 experiment: decode GEMM with four n-tiles per wave on one accumulator chain (dgemm_nt2 = 4) against the product form:
numerics (fp32 re-association only) on the five GEMM shapes of the decode step at 48 / 96 / 192 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L
from shapeformer_amd.gpt import pack_skinny16
lib, dev = L.lib(), torch.device("cuda:0")
for M in (48, 96, 192):
    g = torch.Generator(device="cpu").manual_seed(M)
    Mp = int(lib.sfmi_decode_gemm_padded_rows(M))
    for (N, K, ln, act, use_res, S) in [(3072, 1024, 1, 0, False, 1), (1024, 1024, 0, 0, True, 1), (4096, 1024, 1, 1, False, 1),
                                        (1024, 4096, 0, 0, True, 4), (4097, 1024, 1, 0, False, 1)]:
        Np = (N + 15) // 16 * 16
        wp = pack_skinny16(torch.randn(N, K, generator=g) * 0.05).to(dev)
        x = torch.randn(Mp * K, generator=g).to(dev)
        c1, c2 = torch.randn(Np, generator=g).to(dev), torch.randn(Np, generator=g).to(dev)
        packed = 0 if N == 4097 else 1
        ldo = 4128 if N == 4097 else N
        res = torch.randn(Mp * N, generator=g).to(dev) if use_res else None
        slab = torch.empty(lib.sfmi_decode_gemm_slab_floats(Mp, 4096, 4), device=dev)
        cnt = torch.zeros(Mp // 16 * 260, device=dev, dtype=torch.int32)
        outs = []
        for knob in (1, 4):
            L.check(lib.sfmi_tune_set(b"dgemm_nt2", knob), "tune")
            out = torch.full((Mp * max(N, ldo),), 7.0, device=dev)
            L.check(lib.sfmi_decode_gemm_f32(L.ptr(x), L.ptr(wp), L.ptr(c1) if ln else None, L.ptr(c2), L.ptr(res), L.ptr(out), M, N, K, ldo,
                                             ln, act, packed, S, L.ptr(slab) if S > 1 else None, L.ptr(cnt) if S > 1 else None, L.stream_ptr()), "gemm")
            torch.cuda.synchronize()
            outs.append(out.cpu())
        n = (M + 15) // 16 * (N // 16) * 256 if packed else M * ldo
        a, b = outs[0][:n], outs[1][:n]
        if not packed:      # only the first N columns of each row are written
            a, b = a.view(M, ldo)[:, :N], b.view(M, ldo)[:, :N]
        print(f"M={M} N={N} K={K} ln={ln} act={act} S={S}: max|diff| {float((a - b).abs().max()):.3e} (scale {float(a.abs().max()):.2f}) "
              f"finite {bool(torch.isfinite(b).all())}")
L.check(lib.sfmi_tune_set(b"dgemm_nt2", 1), "tune")
