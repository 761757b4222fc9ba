"""csrc/sgemm_sk.hip (work-balanced GEMM) against csrc/sgemm.hip (+ its split-K reduce launch) on the GEMM shapes of the
transformer training step at batch 1 (M = 499 tokens) and batch 8 (M = 3992), per tile / grid setting.

    python tools/kbench_sk.py [--m 499,3992] [--grids 512] [--tiles 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import ev_time
from shapeformer_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--m", default="499,3992")
ap.add_argument("--grids", default="512")
ap.add_argument("--tiles", default="0")
ap.add_argument("--loops", default="0", help="sk_loop values (64 x 64 form: 0 plain, 1 pipelined + interleaved)")
ap.add_argument("--staggers", default="0", help="sk_stagger values (x 256 cycles start delay of the second workgroup of a CU)")
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
lib = L.lib()
slab = torch.empty(lib.sfmi_sgemm_sk_slab_floats(), device=dev)
cnt = torch.zeros(1 << 20, device=dev, dtype=torch.int32)
D = 1024
tot = {}
for M in (int(x) for x in a.m.split(",")):
    # (name, tA, tB, m, n, k): y = x W^T ; dX = dY W ; dW = dY^T X of the four Linear layers of a block
    shapes = []
    for nm, n_out, k_in in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)):
        shapes += [(nm + ".fwd", 0, 1, M, n_out, k_in), (nm + ".dX", 0, 0, M, k_in, n_out), (nm + ".dW", 1, 0, n_out, k_in, M)]
    print(f"M = {M} tokens")
    for nm, tA, tB, m, n, k in shapes:
        A = torch.randn((k, m) if tA else (m, k), device=dev)
        B = torch.randn((n, k) if tB else (k, n), device=dev)
        Cm = torch.empty(m, n, device=dev)
        ws = torch.empty(max(lib.sfmi_sgemm_mfma_splits(m, n, k), 1) * m * n, device=dev)
        lda, ldb = A.shape[1], B.shape[1]
        f_old = lambda: L.check(lib.sfmi_sgemm_mfma_f32(tA, tB, m, n, k, L.ptr(A), lda, L.ptr(B), ldb, L.ptr(Cm), n, 0, None, 0, None, L.ptr(ws), ws.numel(), 0.0, 0,
                                                        L.stream_ptr()), "sgemm")
        f_new = lambda: L.check(lib.sfmi_sgemm_sk_f32(tA, tB, m, n, k, L.ptr(A), lda, L.ptr(B), ldb, L.ptr(Cm), None, n, 0, None, 0, None, None, 0.0, 0, L.ptr(slab),
                                                      slab.numel(), L.ptr(cnt), cnt.numel(), L.stream_ptr()), "sgemm_sk")
        fl = 2.0 * m * n * k
        t0 = ev_time(f_old, a.reps)
        line = f"  {nm:9s} {m:5d} x {n:5d} x {k:5d}  sgemm S={lib.sfmi_sgemm_mfma_splits(m, n, k)} {t0 * 1e3:7.1f} us {fl / t0 / 1e9:6.1f} TF |"
        tot.setdefault((M, "old"), 0.0)
        tot[(M, "old")] += t0
        for tile in (int(x) for x in a.tiles.split(",")):
            for grid in (int(x) for x in a.grids.split(",")):
                for loop in (int(x) for x in a.loops.split(",")):
                    for stg in (int(x) for x in a.staggers.split(",")):
                        for kname, val in ((b"sk_tile", tile), (b"sk_grid", grid), (b"sk_loop", loop), (b"sk_stagger", stg)):
                            L.check(lib.sfmi_tune_set(kname, val), "tune")
                        t1 = ev_time(f_new, a.reps)
                        line += f" t{tile if tile else lib.sfmi_sgemm_sk_tile(m, n, k)}{'*' if not tile else ''} g{grid} l{loop} s{stg} {t1 * 1e3:6.1f} us {fl / t1 / 1e9:5.1f} TF |"
                        tot.setdefault((M, tile, grid, loop, stg), 0.0)
                        tot[(M, tile, grid, loop, stg)] += t1
        print(line, flush=True)
for kname, val in ((b"sk_tile", 0), (b"sk_grid", 512), (b"sk_loop", 0), (b"sk_stagger", 0)):
    L.check(lib.sfmi_tune_set(kname, val), "tune")
print("sum over the 12 GEMMs of a block (ms):", {str(k): round(v, 3) for k, v in tot.items()})
