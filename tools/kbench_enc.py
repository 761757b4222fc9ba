"""Per-point encoder (sfmi_encode_points_f32: cell grouping + 5 ResnetBlockFC stages with 4 local max pools + scatter_mean) timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L, synthetic
from shapeformer_amd.vqdif import VQDIF
from bench import ev_time
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--fused", type=int, default=1, help="enc_fused knob (csrc/encoder.hip): 1 = the five stages in one launch, 0 = one launch per stage")
a = ap.parse_args()
dev = torch.device("cuda:0")
vq = VQDIF(res=16, device=dev)
lib = L.lib()
L.check(lib.sfmi_tune_set(b"enc_fused", a.fused), "tune")
print(f"enc_fused = {a.fused}")
for B, T in ((64, 16384), (32, 32768), (8, 16384)):
    X = torch.from_numpy(synthetic.make_batch(99, B, n_partial=T, n_full=T)["Xct" if T == 16384 else "Xbd"]).to(dev)
    ws = torch.empty(lib.sfmi_enc_workspace_bytes(B, T), device=dev, dtype=torch.uint8)
    g64 = torch.empty(B, 64, 64, 64, 32, device=dev)
    msk = torch.empty(B, 16, 16, 16, device=dev, dtype=torch.uint8)
    f = lambda: L.check(lib.sfmi_encode_points_f32(L.ptr(X), L.ptr(vq.enc_w), L.ptr(g64), L.ptr(msk), None, L.ptr(ws), B, T, 16, L.stream_ptr()), "enc")
    ms = ev_time(f, 10)
    alg = B * (4 * (2 * T * 128 + T * 4) + T * 128 + 64 ** 3 * 128)
    print(f"B={B} T={T}: {ms:.3f} ms  ({ms / B * 1e3:.1f} us/shape, algorithmic {alg / ms / 1e6:.0f} GB/s = {alg / ms / 1e6 / 8000:.3f} of HBM peak; "
          f"{B * T * 53.6e3 / ms / 1e9:.1f} TFLOP/s)")
    # the product route: the same pipeline with the first Downsampler convolution fused in (no dense grid), and the whole
    # encode_cl (encoder + Downsampler + GroupNorm statistics) on both routes
    d0 = vq.down[0]
    y = torch.empty(B, 32, 32, 32, 64, device=dev)
    f2 = lambda: L.check(lib.sfmi_encode_points_down_f32(L.ptr(X), L.ptr(vq.enc_w), L.ptr(d0.w), L.ptr(y), L.ptr(msk), None, L.ptr(ws), B, T, 16, 1,
                                                          L.stream_ptr()), "enc_down")
    ms2 = ev_time(f2, 10)
    alg2 = B * (4 * (2 * T * 128 + T * 4) + T * 128 + 32 ** 3 * 256)
    t = {}
    for fuse in (True, False):
        vq.FUSE_DOWN0 = fuse
        t[fuse] = ev_time(lambda: vq.encode_cl(X), 5)
    vq.FUSE_DOWN0 = True
    print(f"      fused first Downsampler conv (sfmi_encode_points_down_f32): {ms2:.3f} ms, algorithmic {alg2 / ms2 / 1e6:.0f} GB/s = {alg2 / ms2 / 1e6 / 8000:.3f} of HBM peak; "
          f"encode_cl (encoder + Downsampler + GN): fused {t[True]:.3f} ms, dense route {t[False]:.3f} ms")
