"""encode_cl (64 x 16384 points) for a kernel trace: real clouds, then degenerate clouds (all points in one cell)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import synthetic
from shapeformer_amd.vqdif import VQDIF
dev = torch.device("cuda:0")
vq = VQDIF(res=16, device=dev)
B, T = 64, 16384
X = torch.from_numpy(synthetic.make_batch(99, B, n_partial=T, n_full=T)["Xct"]).to(dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "real"
if mode == "one_cell":
    X = torch.zeros_like(X) + 0.3
lat, mask = vq.encode_cl(X)
print(mode, "occupied 16^3 cells per shape:", float(mask.float().sum() / B))
for _ in range(6):
    vq.encode_cl(X)
torch.cuda.synchronize()
