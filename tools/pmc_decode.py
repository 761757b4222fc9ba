"""Short decode run for rocprofv3 --pmc passes (eager launches, no hipGraph): B rows, prefill + a few steps at a
realistic cached length, so per-launch FETCH_SIZE / WRITE_SIZE of dgemm_kernel / attn_decode_kernel can be read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shapeformer_amd.gpt import CondTupleGPT
B = int(os.environ.get("B", 64)); Lc0 = int(os.environ.get("LC", 400)); steps = int(os.environ.get("STEPS", 4))
dev = torch.device("cuda:0")
g = CondTupleGPT(device=dev)
rs = np.random.RandomState(0)
c = np.full((B, Lc0, 2), 4096, np.int64)
for b in range(B):
    c[b, :Lc0 - 1, 0] = np.sort(rs.choice(4096, Lc0 - 1, replace=False)); c[b, :Lc0 - 1, 1] = rs.randint(0, 4096, Lc0 - 1)
out = g.sample(torch.from_numpy(c), torch.full((B,), Lc0, dtype=torch.int32), max_steps=steps, stop_early=False, use_graph=False)
torch.cuda.synchronize()
print("done", out["steps"])
