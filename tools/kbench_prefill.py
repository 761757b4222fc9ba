"""attn_prefill_mfma_kernel alone: rows x 16 heads x P positions, causal; useful-FLOP fraction of the f32 MFMA peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L
from bench import ev_time
dev = torch.device("cuda:0")
lib = L.lib()
D, H, Lmax = 1024, 16, 813
for B, P in ((64, 160), (96, 160), (96, 300), (16, 400), (8, 811)):
    qkv = torch.randn(B * P, 3 * D, device=dev)
    y = torch.empty(B * P, D, device=dev)
    nv = torch.full((B,), P, device=dev, dtype=torch.int32)
    Kc, Vc = torch.empty(B, Lmax, D, device=dev), torch.empty(B, Lmax, D, device=dev)
    f = lambda: L.check(lib.sfmi_gpt_attn_prefill_f32(L.ptr(qkv), L.ptr(Kc), L.ptr(Vc), L.ptr(nv), L.ptr(y), B, P, D, H, Lmax, None, 0.0, 0,
                                                      L.stream_ptr()), "attn_prefill")
    ms = ev_time(f, 20)
    fl = B * H * (P * (P + 1) / 2) * 4 * 64
    print(f"B={B:3d} P={P:3d}: {ms * 1e3:8.1f} us  useful {fl / ms / 1e9:6.1f} TFLOP/s = {fl / ms / 1e9 / 157.3:.3f} of f32 MFMA")
