"""Micro-benchmark of the SDF-query kernel (run on the GPU box)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import ops, weights as W

dev = torch.device("cuda:0")
B, Q = int(os.environ.get("B", 16)), int(os.environ.get("Q", 128))
sd = W.make_state_dict(W.vqdif_spec(16))
wp = torch.from_numpy(ops.sdf_pack_weights(sd)).to(dev)
grid = torch.randn(B, 64, 64, 64, 32, device=dev)
axis = torch.linspace(-1, 1, Q, device=dev)
pts = torch.rand(B, Q ** 3, 3, device=dev) * 2 - 1
out = torch.empty(B, Q ** 3, 1, device=dev)
for name, fn in (("grid", lambda: ops.sdf_query_grid(axis, grid, wp, out=out)),
                 ("points", lambda: ops.sdf_query(pts, grid, wp, out=out))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    npts = B * Q ** 3
    print(json.dumps({"kernel": "sdf_query_" + name, "B": B, "Q": Q, "ms": ms, "Gpts_s": npts / ms / 1e6,
                      "TFLOPs": npts * 31488 / ms / 1e9, "frac_f32_peak": npts * 31488 / ms / 1e9 / 157.3}))
