"""Per-layer timing of the VQDIF decoder's feature-grid path (UNet3D + Upsampler) at B shapes: every conv launch group,
GroupNorm statistics, pooling / concat, final affine.  FLOPs as executed (sub-pixel up-sampling convs: 8 taps).
GPU box only:  python tools/kbench_conv.py [--batch 64]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd.vqdif import VQDIF

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--xreuse", type=int, default=2, help="conv_xreuse knob (csrc/conv3d.hip): 2 = x reuse for every tile width + 128 x 64 tiles on coarse grids, 1 = round-4 form (32 / 64 channels), 0 = per-tap staging")
a = ap.parse_args()
from shapeformer_amd import _lib as L
L.check(L.lib().sfmi_tune_set(b"conv_xreuse", a.xreuse), "tune")
dev = torch.device("cuda:0")
vq = VQDIF(res=16, device=dev)
code = vq.get_code_cl(torch.randint(0, vq.K, (a.batch, 16, 16, 16), device=dev)).clone()
rec = collections.OrderedDict()


def timed(fn, label, flops_of):
    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        rec.setdefault(label(*args, **kw), []).append((e0, e1, flops_of(out, *args, **kw)))
        return out
    return w


def conv_flops(y, x, cv, name, scale=None, shift=None, up=0, relu=True, bias=None, stats=False):
    y = y[0] if stats else y          # _conv(stats=True) -> (y, splits)
    taps = 8 if (up and cv.w_up is not None) else cv.ks ** 3
    return 2.0 * y.numel() // cv.cout * cv.cout * cv.cin * taps


dec = vq.decoder            # the LocalDecoder sub-module owns the grid ops since round 5
dec._conv = timed(dec._conv, lambda x, cv, name, *r, **k: f"conv {name:16s} {tuple(x.shape[1:4])}x{cv.cin}->{cv.cout} k{cv.ks}" + (" up2" if k.get("up") else ""), conv_flops)
dec._gn = timed(dec._gn, lambda x, g, b, name, partial_S=None: f"gn   {name}" + (" (coefficients from the conv's partials)" if partial_S else ""), lambda *r, **k: 0.0)
dec._pool = timed(dec._pool, lambda x, name: f"pool {name}", lambda *r, **k: 0.0)
dec._upcat = timed(dec._upcat, lambda s, l, name: f"cat  {name}", lambda *r, **k: 0.0)
dec._affine = timed(dec._affine, lambda x, sc, sh, name: f"aff  {name}", lambda *r, **k: 0.0)
for _ in range(2 + a.reps):
    vq.decoder_grid_cl(code)
torch.cuda.synchronize()
tot = 0.0
for k, v in rec.items():
    ms = sum(e0.elapsed_time(e1) for e0, e1, _ in v[2:]) / len(v[2:])
    tot += ms
    fl = v[-1][2]
    print(f"{k:60s} {ms:8.3f} ms" + (f"  {fl / ms / 1e9:7.1f} TFLOP/s  ({fl / 1e9 / a.batch:6.2f} GF/shape)" if fl else ""))
print(f"total {tot:.2f} ms for {a.batch} shapes (conv_xreuse={a.xreuse})")
