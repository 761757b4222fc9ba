"""How many of the decode chains' streams really run at the same time?  k single-wavefront 200 us spins on k probed streams
(gpt._chain_streams): k-way concurrency gives ~0.23 ms for every k.  Also with 2 ms of spins split into 100 short kernels per
stream (dispatch-rate view).   GPU box only:  python tools/probe_queue_concurrency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L, weights as W
from shapeformer_amd.gpt import CondTupleGPT

dev = torch.device("cuda:0")
sd = W.make_state_dict(W.gpt_spec(n_embd=128, n_layers=(2, 1), block_size=96))
g = CondTupleGPT(sd, n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
S = g._chain_streams(4)
cur = torch.cuda.current_stream()
spin = lambda s, t: L.check(L.lib().sfmi_stream_spin(t, s.cuda_stream), "spin")


def run(k, ticks, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for s in S[:k]:
        s.wait_event(e0)
    for _ in range(n):
        for s in S[:k]:
            spin(s, ticks)
    for s in S[:k]:
        cur.wait_stream(s)
    e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1)


run(4, 100, 4)
for k in (1, 2, 3, 4):
    print(f"{k} streams: one 200 us spin each {run(k, 20000, 1):.3f} ms; 100 x 10 us spins each {run(k, 1000, 100):.3f} ms; "
          f"400 x 1 us spins each {run(k, 100, 400):.3f} ms")
