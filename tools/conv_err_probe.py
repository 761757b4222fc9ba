"""End-to-end logit error of the VQDIF decoder against the reference fixture (tests/golden/vqdif16_small.npz) per conv_xreuse form:
max of |d| / (2e-4 + 1e-4 |y|) (SURVEY App.B gate: <= 1) and max |d|, for the fixture's 2 shapes alone and for the same 2 shapes
inside a batch of 64 (other tile / dispatch choices), plus the distance between the forms themselves (this implementation's own
summation-order noise).  GPU box only:  python tools/conv_err_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from shapeformer_amd import _lib as L, weights as W
from shapeformer_amd.vqdif import VQDIF

z = np.load(os.path.join(ROOT, "tests", "golden", "vqdif16_small.npz"))
dev = torch.device("cuda:0")
vq = VQDIF(W.make_state_dict(W.vqdif_spec(16)), res=16, device=dev)
q = torch.from_numpy(z["quant_ind"].astype(np.int64))
Q = int(z["Q"])
ref = z["logits"]
tol = 2e-4 + 1e-4 * np.abs(ref)
got = {}
for knob in (0, 1, 2, 3):
    L.check(L.lib().sfmi_tune_set(b"conv_xreuse", knob), "tune")
    for nb in (2, 64):
        qq = q if nb == 2 else torch.cat([q, torch.randint(0, vq.K, (nb - 2,) + tuple(q.shape[1:]))])
        lg = vq.decode_index(qq, grid_Q=Q)["logits"].cpu().numpy()[:2, ..., 0].astype(np.float64)
        got[(knob, nb)] = lg
        d = np.abs(lg - ref)
        print(f"conv_xreuse {knob} batch {nb:2d}: max |d| {d.max():.3e}  max |d|/tol {(d / tol).max():.3f}  rms {np.sqrt((d ** 2).mean()):.3e}  over-gate {(d > tol).sum()} / {d.size}")
ks = sorted(got)
for i, a in enumerate(ks):
    for b in ks[i + 1:]:
        d = np.abs(got[a] - got[b])
        print(f"{a} vs {b}: max |d| {d.max():.3e} rms {np.sqrt((d ** 2).mean()):.3e}")
