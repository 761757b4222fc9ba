"""Which HIP streams should the interleaved decode chains run on?  Creates 8 streams (first use in order), prints the
pairwise spin-overlap probe of gpt._chain_streams, then times the real 192-shape AR loop (3 chains) on several stream triples.
GPU box only:  python tools/probe_chain_streams.py [--steps 512]"""
import argparse, itertools, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L, synthetic
from shapeformer_amd.gpt import CondTupleGPT
from shapeformer_amd.pipeline import ShapeCompletion
from shapeformer_amd.vqdif import VQDIF

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=512)
ap.add_argument("--batch", type=int, default=192)
ap.add_argument("--combos", default="0,1,2;1,2,3;4,5,6;0,1,2")
ap.add_argument("--serial-prefill", default="0,1,0,1", help="per combo: 1 = prefills issued one after another on the caller's stream")
a = ap.parse_args()
dev = torch.device("cuda:0")
vq = VQDIF(res=16, device=dev)
gpt = CondTupleGPT(device=dev)
pipe = ShapeCompletion(vq, gpt)
kept, seed = [], 314
while sum(k.shape[0] for k in kept) < a.batch:
    cand = torch.from_numpy(synthetic.make_batch(seed, 48, n_partial=16384)["Xct"]).to(dev)
    seed += 48
    lcs = pipe.encode_cloud(cand)["Lc"].clone()
    kept.append(cand[torch.nonzero(lcs <= gpt.Lmax - a.steps).flatten()])
enc = pipe.encode_cloud(torch.cat(kept)[:a.batch].contiguous())
S = [torch.cuda.Stream(device=dev) for _ in range(8)]
spin = lambda s, t: L.check(L.lib().sfmi_stream_spin(t, s.cuda_stream), "spin")
for s in S:
    spin(s, 1); s.synchronize()
cur = torch.cuda.current_stream()

def pair(x, y, ticks=20000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for s in (x, y):
        s.wait_event(e0); spin(s, ticks)
    for s in (x, y):
        cur.wait_stream(s)
    e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1)

print("pairwise 200 us spins (ms):")
for i in range(8):
    print("  ", " ".join(f"{pair(S[i], S[j]):.2f}" if j > i else "  - " for j in range(8)))
kw = dict(max_steps=a.steps, top_k=100, top_p=0.4, temperature=1.0, best_in_first=False, mask_invalid=True,
          mask_invalid_completion=True, stop_early=False)
for combo, ser in zip(a.combos.split(";"), a.serial_prefill.split(",")):
    idx = [int(c) for c in combo.split(",")]
    gpt.PREFILL_ON_CHAIN_STREAMS = ser != "1"
    gpt._mb_streams = [S[i] for i in idx]
    out = []
    for rep in range(2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        gpt.sample_microbatched(enc["c_tokens"], enc["Lc"], n_micro=3, seed=rep, after_prefill=lambda: ev[1].record(), **kw)
        ev[2].record(); torch.cuda.synchronize()
        out.append((round(ev[0].elapsed_time(ev[1]), 1), round(ev[1].elapsed_time(ev[2]) / a.steps, 3)))
    print(json.dumps({"streams": idx, "serial_prefill": ser == "1", "prefill_ms,ar_ms_per_step": out}), flush=True)
