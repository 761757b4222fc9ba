"""Phase-by-phase repro of the lock-step multi-stream decode graph on a tiny model (faulthandler on)."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shapeformer_amd import weights as W
from shapeformer_amd.gpt import CondTupleGPT
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ar_sweep import synth_cond
dev = torch.device("cuda:0")
kw = dict(n_embd=128, n_layers=(2, 1), block_size=400)
g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=400, device=dev)
tok, Lc = synth_cond(3, 32, lo=20, hi=40, Lpad=200)
def say(*a): print(*a, flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
ref = g.sample_microbatched(tok, Lc, n_micro=2, max_steps=8, stop_early=False)
ref_seq = ref["state"]["seq"].clone()
say("reference (independent chains) done")
if mode == "events":      # minimal two-stream capture with an event edge, no sfmi kernels
    s1, s2, cap = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1024, device=dev); y = torch.zeros(1024, device=dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=cap):
        s1.wait_stream(cap); s2.wait_stream(cap)
        with torch.cuda.stream(s1):
            x.add_(1); e = torch.cuda.Event(); e.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(e); y.add_(x)
        cap.wait_stream(s1); cap.wait_stream(s2)
    say("captured"); gr.replay(); torch.cuda.synchronize(); say("replayed", float(x[0]), float(y[0]))
    sys.exit(0)
g.ROT_LANES, g.ROT_STEPS = 1, 1
orig_steps, orig_graph = g._rot_steps, g._rot_graph
def steps_dbg(*a, **k):
    say("  _rot_steps enter"); r = orig_steps(*a, **k); say("  _rot_steps exit"); return r
g._rot_steps = steps_dbg
def graph_dbg(*a, **k):
    say(" _rot_graph enter"); r = orig_graph(*a, **k); torch.cuda.synchronize(); say(" _rot_graph exit (captured + instantiated)"); return r
g._rot_graph = graph_dbg
out = g.sample_microbatched(tok, Lc, n_micro=2, max_steps=8, stop_early=False)
torch.cuda.synchronize()
say("lock-step run done; tokens equal to the independent-chain run:", bool(torch.equal(out["state"]["seq"], ref_seq)))
