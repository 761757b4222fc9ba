"""BASELINE config 2 alone (VQDIF-16 reconstruction of 32 clouds x 32768 points on the 64^3 lattice: `pipeline.reconstruct`): wall time per
call next to what a kernel trace of the same process shows (tools/gpu_call.sh ... "trace=config2:python /root/repo/tools/prof_config2.py")."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import synthetic
from shapeformer_amd.pipeline import ShapeCompletion
from shapeformer_amd.vqdif import VQDIF

dev = torch.device("cuda:0")
vq = VQDIF(res=16, device=dev)
X = torch.from_numpy(synthetic.make_batch(2000, 32)["Xbd"]).to(dev)
pipe = ShapeCompletion(vq, None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = pipe.reconstruct(X, decode_res=64, max_length=512)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"call {i}: {dt * 1e3:.2f} ms  ({32 / dt:.0f} shapes/s)", flush=True)
