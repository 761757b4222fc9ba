"""Short run for rocprofv3 --pmc passes of the implicit-decoder kernel and the attention kernels: 4 x 128^3 SDF grid
queries at B shapes, one prefill at a realistic prefix and a few decode steps (eager launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shapeformer_amd import ops, weights as W
from shapeformer_amd.vqdif import VQDIF
B = int(os.environ.get("B", 8))
dev = torch.device("cuda:0")
vq = VQDIF(res=16, device=dev)
grid = torch.randn(B, 64, 64, 64, 32, device=dev)
axis = torch.from_numpy(np.linspace(-1.0, 1.0, 128).astype(np.float32)).to(dev)
for _ in range(4):
    out = ops.sdf_query_grid(axis, grid, vq.sdf_w, sigmoid=True)
torch.cuda.synchronize()
print("sdf done", tuple(out.shape))
