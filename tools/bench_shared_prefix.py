"""BASELINE config 3's real shape: sample_n = 16 completions of ONE shape, 512 steps - expanded rows vs shared prefix."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from shapeformer_amd.gpt import CondTupleGPT
dev = torch.device("cuda:0")
g = CondTupleGPT(device=dev)
rs = np.random.RandomState(0)
for S, Lc in ((16, 40), (16, 84), (16, 120), (16, 150), (16, 200), (16, 300), (16, 400), (64, 84), (64, 150), (64, 300)):
    c = np.full((1, Lc, 2), 4096, np.int64)
    c[0, :Lc - 1, 0] = np.sort(rs.choice(4096, Lc - 1, replace=False)); c[0, :Lc - 1, 1] = rs.randint(0, 4096, Lc - 1)
    ct = torch.from_numpy(c).expand(S, -1, -1).contiguous().to(dev, torch.int32)
    Lt = torch.full((S,), Lc, dtype=torch.int32, device=dev)
    for shared in (False, True):
        for it in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            g.sample(ct, Lt, max_steps=512, seed=it, stop_early=False, to_host=False, shared_prefix=shared)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"S={S} Lc={Lc} shared_prefix={shared}: {dt * 1e3:.1f} ms for 512 steps ({dt / 512 * 1e3:.3f} ms/step)", flush=True)
