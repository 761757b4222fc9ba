"""Which cross-stream dependency patterns survive hipStreamEndCapture + hipGraphInstantiate (torch ops only)?"""
import faulthandler, sys
faulthandler.enable()
import torch
dev = torch.device("cuda:0")
pat = sys.argv[1]
s1, s2, cap = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 16, device=dev); y = torch.zeros(1 << 16, device=dev)
x.add_(0); y.add_(0); torch.cuda.synchronize()
keep = []
def ev(s):
    e = torch.cuda.Event(); e.record(s); keep.append(e); return e
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr, stream=cap):
    s1.wait_stream(cap); s2.wait_stream(cap)
    if pat == "zigzag":            # s1 -> s2 -> s1 -> s2 ...
        e = None
        for r in range(4):
            with torch.cuda.stream(s1):
                x.add_(1)
                if e is not None: s1.wait_event(e)
                x.mul_(1.5); e = ev(s1); x.add_(2)
            with torch.cuda.stream(s2):
                y.add_(1); s2.wait_event(e); y.mul_(1.5); e = ev(s2); y.add_(2)
    elif pat == "oneway":          # only s1 -> s2 edges, several
        for r in range(4):
            with torch.cuda.stream(s1):
                x.add_(1); e = ev(s1); x.add_(2)
            with torch.cuda.stream(s2):
                y.add_(1); s2.wait_event(e); y.add_(2)
    elif pat == "pingpong_tail":   # the event is the LAST thing recorded on a stream before the other waits (no trailing work)
        e = None
        for r in range(4):
            with torch.cuda.stream(s1):
                if e is not None: s1.wait_event(e)
                x.add_(1); e = ev(s1)
            with torch.cuda.stream(s2):
                s2.wait_event(e); y.add_(1); e = ev(s2)
    cap.wait_stream(s1); cap.wait_stream(s2)
print(pat, "captured + instantiated", flush=True)
for _ in range(3): gr.replay()
torch.cuda.synchronize()
print(pat, "replayed", float(x[0]), float(y[0]), flush=True)
