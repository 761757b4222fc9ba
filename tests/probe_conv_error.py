"""Where the fp32 noise of the HIP decoder comes from (not a pytest file; GPU box:  python tests/probe_conv_error.py).

The float64 evaluation of the oracle is the rounding-free value.  Per UNet3D layer (GroupNorm -> conv3 -> ReLU), on the float64
layer INPUT rounded to fp32:  rms error relative to rms(output) of (a) torch-CPU fp32, (b) the HIP layer per conv_xreuse form.
Then the whole chain: HIP grid vs float64 grid, float64 SDF query on the HIP grid vs float64 logits (conv-stack share), HIP logits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from oracle import vqdif_oracle as O
from shapeformer_amd import _lib as L, weights as W
from shapeformer_amd.vqdif import VQDIF

torch.set_num_threads(16)
z = np.load(os.path.join(ROOT, "tests", "golden", "vqdif16_small.npz"))
dev = torch.device("cuda:0")
sd_np = W.make_state_dict(W.vqdif_spec(16))
sd = O.to_torch_sd(sd_np)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
vq = VQDIF(sd_np, res=16, device=dev)
dec = vq.decoder
q = torch.from_numpy(z["quant_ind"].astype(np.int64))
lib = L.lib()


def rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())


def layer(x64, name, sc):
    pre = f"decoder.unet3d.{name}.basic_module.{sc}."
    y64 = O.single_gcr(sd64, pre, x64)
    y32 = O.single_gcr(sd, pre, x64.float())
    xcl = x64.float().permute(0, 2, 3, 4, 1).contiguous().to(dev)
    out = [f"{name}.{sc:12s} K={27 * x64.shape[1]:6d}  torch-cpu fp32 {rel(y32, y64):.2e}"]
    for knob in (0, 2, 3):
        L.check(lib.sfmi_tune_set(b"conv_xreuse", knob), "tune")
        y = dec._single_gcr(xcl, dec.unet[f"{name}.{sc}"], "probe").cpu().permute(0, 4, 1, 2, 3)
        out.append(f"hip form {knob} {rel(y, y64):.2e}")
    print("   ".join(out), flush=True)
    return y64


with torch.no_grad():
    x = O.get_code(sd64, q)
    e0 = layer(layer(x, "encoders.0", "SingleConv1"), "encoders.0", "SingleConv2")
    e1 = layer(layer(F.max_pool3d(e0, 2), "encoders.1", "SingleConv1"), "encoders.1", "SingleConv2")
    e2 = layer(layer(F.max_pool3d(e1, 2), "encoders.2", "SingleConv1"), "encoders.2", "SingleConv2")
    y = torch.cat([e1, F.interpolate(e2, size=e1.shape[2:], mode="nearest")], dim=1)
    y = layer(layer(y, "decoders.0", "SingleConv1"), "decoders.0", "SingleConv2")
    y = torch.cat([e0, F.interpolate(y, size=e0.shape[2:], mode="nearest")], dim=1)
    y = layer(layer(y, "decoders.1", "SingleConv1"), "decoders.1", "SingleConv2")
    # whole chain
    g64 = O.decoder_grid(sd64, O.get_code(sd64, q))
    g32 = O.decoder_grid(sd, O.get_code(sd, q))
    Q = int(z["Q"])
    Xtg = torch.from_numpy(O.make_grid(Q, dtype=np.float64))[None].expand(2, -1, -1)
    l64 = O.sdf_query(sd64, g64, Xtg)[..., 0]
    gate = 2e-4 + 1e-4 * l64.abs()
    print(f"grid: torch-cpu fp32 {rel(g32, g64):.2e}")
    for knob in (0, 1, 2, 3):
        L.check(lib.sfmi_tune_set(b"conv_xreuse", knob), "tune")
        g = vq.decoder_grid_cl(vq.get_code_cl(q)).cpu().permute(0, 4, 1, 2, 3)
        lq = O.sdf_query(sd64, g.double(), Xtg)[..., 0]            # float64 query of the HIP grid: the conv stack's share
        lg = vq.decode_index(q, grid_Q=Q)["logits"].cpu()[..., 0].double()
        d1, d2 = (lq - l64).abs(), (lg - l64).abs()
        print(f"form {knob}: grid {rel(g, g64):.2e}   logits, conv stack only: rms {float((d1 ** 2).mean().sqrt()):.2e} max/gate {float((d1 / gate).max()):.3f}"
              f"   HIP end to end: rms {float((d2 ** 2).mean().sqrt()):.2e} max/gate {float((d2 / gate).max()):.3f}")
    lq = O.sdf_query(sd64, g32.double(), Xtg)[..., 0]
    d1 = (lq - l64).abs()
    print(f"torch-cpu fp32 grid, float64 query: rms {float((d1 ** 2).mean().sqrt()):.2e} max/gate {float((d1 / gate).max()):.3f}")
