"""Decode-GEMM micro-benchmark (graph of 24 layer launches); SFMI_DGEMM_DBG ablations: 1 no x loads, 2 no MFMA."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd.gpt import CondTupleGPT
from bench import ev_time
B = int(os.environ.get("B", 16))
dev = torch.device("cuda:0")
gpt = CondTupleGPT(device=dev)
st = gpt._alloc(B, 512)
D = gpt.D
r = st["resid"]
for nm, attr, c1a, c2a, xin, res, outb, N, K, ldo, ln, act, S in (
        ("fc1", "pfc1", "c1fc1", "c2fc1", r, None, st["h"], 4 * D, D, 4 * D, 1, 1, 1),
        ("fc1-noln-noact", "pfc1", None, "c2fc1", r, None, st["h"], 4 * D, D, 4 * D, 0, 0, 1),
        ("fc2 S4", "pfc2", None, "bfc2", st["h"], r, r, D, 4 * D, D, 0, 0, 4),
        ("fc2 S1", "pfc2", None, "bfc2", st["h"], r, r, D, 4 * D, D, 0, 0, 1),
        ("fc2 S2", "pfc2", None, "bfc2", st["h"], r, r, D, 4 * D, D, 0, 0, 2),
        ("qkv", "pqkv", "c1qkv", "c2qkv", r, None, st["qkv"], 3 * D, D, 3 * D, 1, 0, 1),
        ("proj S4", "pproj", None, "bproj", st["y"], r, r, D, D, D, 0, 0, 4),
        ("proj S2", "pproj", None, "bproj", st["y"], r, r, D, D, D, 0, 0, 2),
        ("proj S1", "pproj", None, "bproj", st["y"], r, r, D, D, D, 0, 0, 1)):
    def body():
        for l in gpt.layers:
            gpt._dgemm(xin, getattr(l, attr), getattr(l, c1a) if c1a else None, getattr(l, c2a), res, outb, B, N, K, ldo, ln, act, 1, S)
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    ms = ev_time(g.replay, 10) / len(gpt.layers)
    print(f"dbg={os.environ.get('SFMI_DGEMM_DBG','0')} B={B} {nm:16s} {ms*1e3:7.2f} us  {N*K*4/ms/1e6:7.1f} GB/s")
