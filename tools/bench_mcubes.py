import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_mcubes_cpu import _sphere, _torus
from shapeformer_amd import mcubes
occ = torch.from_numpy(np.stack([_sphere(128, 0.6), _torus(128)] * 8).astype(np.float32)).cuda()
for _ in range(2): v, f, vo, to = mcubes.marching_cubes_dev(occ, 0.5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): v, f, vo, to = mcubes.marching_cubes_dev(occ, 0.5)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"marching cubes 16 x 128^3: {dt*1e3:.2f} ms/batch = {dt/16*1e3:.3f} ms/shape, {len(v)} verts {len(f)} tris; grid read {16*128**3*4/dt/1e9:.0f} GB/s")
