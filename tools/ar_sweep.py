#!/usr/bin/env python3
"""AR-loop-only A/B of decode-step launch shapes on ONE process (model built once): for each configuration line the decode
graphs are re-captured and the 512-step loop of `--rows` synthetic sequences (`--chains` interleaved hipGraph chains) is timed
with HIP events, optionally with one kernel family disabled (timing-only ablation, gpt._ablate).

    python tools/ar_sweep.py --out gpurun_out/r3/ar_sweep.txt [--rows 320 --chains 4] < configs

A configuration line is `name key=value ...`; keys: the sfmi_tune_set knobs (attn_blocks, attn_unroll, attn_waves,
attn_lds_pad, ...), `ablate=gemm|attn`, `rows=`, `chains=`, `lanes=` (gpt.ATTN_LANES: attention turnstile, at most that many
chains stream their KV cache at a time), `prefetch=` (gpt.PREFETCH_BLOCKS: Infinity-Cache weight prefetch branch of a single chain),
`profile=attn,gemm` (in-situ launch timing: prints launches and mean us per family), `cus=<n>` (every chain on a stream restricted to
the first n compute-unit mask bits = n / 8 CUs of every XCD; hipExtStreamCreateWithCUMask), `cusplit=<n>` (chains 0, 1 on the first n
bits, chains 2, 3 on the other 256 - n: spatial partition of the two kernel families together with a per-chain ablate=),
`cumode=block` (mask bits taken as contiguous blocks instead: bits [0, n) vs [n, 256)), `hwid=1` (print which XCC / SE / CU the masked
streams' workgroups land on).  Lines starting with # are skipped.
Condition lengths are uniform in [100, 216] (mean 158 = the bench's synthetic clouds); positions ascending, end-token closed.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def synth_cond(seed, B, lo=100, hi=216, Lpad=406):
    rs = np.random.RandomState(seed)
    tok = np.full((B, Lpad, 2), 4096, np.int32)
    Lc = rs.randint(lo, hi + 1, B).astype(np.int32)
    for b in range(B):
        n = Lc[b] - 1
        tok[b, :n, 0] = np.sort(rs.choice(4096, n, replace=False))
        tok[b, :n, 1] = rs.randint(0, 4096, n)
    return torch.from_numpy(tok), torch.from_numpy(Lc)


class PowerProbe:
    """Samples the GPU's package power and shader clock from sysfs (hwmon) in a thread while a configuration runs: is the loop
    power / clock limited when HBM-bound and MFMA-bound kernels run together?"""

    def __init__(self):
        import glob
        hw = []
        try:        # the hwmon node of THE device this process computes on (a box shows several cards; only one is ours)
            pr = torch.cuda.get_device_properties(0)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hw = sorted(glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*"))
            self.bdf = bdf
        except Exception as e:
            self.bdf = f"? ({e})"
        if not hw:
            hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.files = {}
        for h in hw:
            for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)), ("mclk_Hz", ("freq2_input",)),
                               ("temp_mC", ("temp1_input", "temp2_input")), ("cap_uW", ("power1_cap",))):
                for n in names:
                    f = os.path.join(h, n)
                    if key not in self.files and os.path.exists(f):
                        self.files[key] = f
            if self.files:
                break
        self.samples, self._stop, self._th = [], False, None

    def _read(self):
        out = {}
        for k, f in self.files.items():
            try:
                out[k] = float(open(f).read().strip())
            except Exception:
                pass
        return out

    def start(self):
        import threading
        self.samples, self._stop = [], False

        def run():
            while not self._stop:
                self.samples.append(self._read())
                time.sleep(0.02)
        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join()
        if not self.samples or not self.files:
            return ""
        def mean(k, sc):
            v = [s[k] for s in self.samples if k in s]
            return f"{sum(v) / len(v) / sc:.0f}" if v else "-"
        def mx(k, sc):
            v = [s[k] for s in self.samples if k in s]
            return f"{max(v) / sc:.0f}" if v else "-"
        return (f"   power {mean('power_uW', 1e6)} W (max {mx('power_uW', 1e6)}, cap {mx('cap_uW', 1e6)}), sclk {mean('sclk_Hz', 1e6)} MHz, mclk {mean('mclk_Hz', 1e6)} MHz, "
                f"temp {mean('temp_mC', 1e3)} C [{len(self.samples)} samples]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r4/ar_sweep.txt")
    ap.add_argument("--rows", type=int, default=320)
    ap.add_argument("--chains", type=int, default=4)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    from shapeformer_amd import _lib as L
    from shapeformer_amd.gpt import CondTupleGPT
    dev = torch.device("cuda:0")
    t0 = time.time()
    gpt = CondTupleGPT(device=dev)
    lib = L.lib()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    out = open(a.out, "a")

    def say(s):
        print(s, flush=True)
        out.write(s + "\n"); out.flush()
    say(f"# ar_sweep: model built in {time.time() - t0:.1f}s; default rows {a.rows} chains {a.chains} steps {a.steps}")
    defaults = {k: lib.sfmi_tune_get(k.encode()) for k in ("attn_blocks", "attn_unroll", "attn_waves", "attn_lds_pad", "sdf_blocks", "dgemm_nt2",
                                                           "dgemm_nw", "dgemm_un")}
    probe = PowerProbe()
    say(f"# power probe: device {probe.bdf}, files {probe.files}")
    bgst = {}        # background SDF-query load (`bgsdf=<shapes per launch>[:<launches>]`): the MFMA-bound decode stage of a previous batch

    def bg_setup(nshape):
        if "vq" not in bgst:
            from shapeformer_amd.vqdif import VQDIF
            bgst["vq"] = VQDIF(res=16, device=dev)
            # a stream that shares NO hardware queue with the chains (probed like the chain streams; needs GPU_MAX_HW_QUEUES > chains)
            ss = gpt._chain_streams(a.chains + 1)
            bgst["stream"] = ss[a.chains]
            say(f"# background stream: probed set of {len(ss)} streams, shared_queue={getattr(gpt, '_mb_shared_queue', False)}, probe ms {getattr(gpt, '_chain_probe', [])[-12:]}")
            bgst["axis"] = torch.linspace(-1, 1, 128, device=dev)
        if bgst.get("n") != nshape:
            bgst["grid"] = torch.randn(nshape, 64, 64, 64, 32, device=dev)
            bgst["out"] = torch.empty(nshape, 128 ** 3, 1, device=dev)
            bgst["n"] = nshape

    def bg_launch(k):
        from shapeformer_amd import ops
        evs = []
        with torch.cuda.stream(bgst["stream"]):
            for _ in range(k):
                ops.sdf_query_grid(bgst["axis"], bgst["grid"], bgst["vq"].sdf_w, sigmoid=True, out=bgst["out"])
                e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        return evs
    import ctypes
    mask_streams = {}

    def hwid_report(stream, tag):
        out_ = torch.zeros(2 * 2048, device=dev, dtype=torch.int32)
        L.check(lib.sfmi_hwid_probe(out_.data_ptr(), 2048, 256, 20000, stream.cuda_stream), "sfmi_hwid_probe")
        stream.synchronize()
        v = out_.cpu().numpy().astype(np.uint32).reshape(-1, 2)
        xcc, hw = v[:, 0] & 0xf, v[:, 1]
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        ids = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
        per_x = {x: sum(1 for i in ids if i[0] == x) for x in sorted(set(xcc.tolist()))}
        say(f"#   hwid {tag}: {len(ids)} distinct (xcc, se, sh, cu); per XCC {per_x}; SE ids {sorted(set(se.tolist()))} CU ids {sorted(set(cu.tolist()))}")
    cache = {}
    for line in sys.stdin:
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        name, *kvs = line.split()
        kv = dict(x.split("=", 1) for x in kvs)
        rows, chains = int(kv.pop("rows", a.rows)), int(kv.pop("chains", a.chains))
        lclo, lchi = int(kv.pop("lclo", 100)), int(kv.pop("lchi", 216))     # condition lengths (short traces: start near the mid-run length)
        gpt._ablate = kv.pop("ablate", "")
        gpt.ATTN_LANES = int(kv.pop("lanes", "0"))
        gpt.S_PROJ_M, gpt.S_FC2 = int(kv.pop("sproj", "1")), int(kv.pop("sfc2", "4"))      # in-kernel split-K of proj / fc2 (part of the graph key)
        gpt._profile = kv.pop("profile", "")
        if kv.pop("part", "0") != "0":
            raise ValueError("part=: the self-placing attention launch was removed from the library after round 6's measurements (profiles/r06_overlap.md; code from commit ac5c852 on)")
        bg = kv.pop("bgsdf", None)
        cus, cusplit, cumode, hwid = int(kv.pop("cus", "0")), int(kv.pop("cusplit", "0")), kv.pop("cumode", "interleave"), int(kv.pop("hwid", "0"))
        if cus or cusplit:
            # masked streams, one per chain (every CU-masked stream owns a hardware queue of its own: ROCclr never pools them); chain i of
            # a split takes side i // 2.  The previous configuration's masked streams are destroyed first - call 1 of round 6 left ~100 of
            # them alive and the later configurations ran on an oversubscribed queue set - and gpt._chain_streams is bypassed: its
            # overlap probe replaced the masked streams by plain ones.
            n_a = cus or cusplit
            sides = [tuple(range(n_a)), tuple(range(n_a, 256))]
            if cumode == "xcdblock":      # whole XCDs: bit b belongs to XCD b % 8 under the interleaved map
                nx = n_a // 32
                sides = [tuple(b for b in range(256) if b % 8 < nx), tuple(b for b in range(256) if b % 8 >= nx)]
            chain_bits = [sides[0] if (cus or i < chains // 2) else sides[1] for i in range(chains)]
            key_ = tuple(chain_bits)
            if mask_streams.get("key") != key_:
                torch.cuda.synchronize()
                for st_ in mask_streams.get("streams", []):
                    L.check(lib.sfmi_stream_destroy(st_.cuda_stream), "sfmi_stream_destroy")
                streams_ = []
                for b in chain_bits:
                    words = (ctypes.c_uint * 8)()
                    for bit in b:
                        words[bit // 32] |= 1 << (bit % 32)
                    sp_ = ctypes.c_void_p()
                    L.check(lib.sfmi_stream_create_cumask(words, 8, ctypes.byref(sp_)), "sfmi_stream_create_cumask")
                    streams_.append(torch.cuda.ExternalStream(sp_.value, device=dev))
                mask_streams["key"], mask_streams["streams"] = key_, streams_
            streams_ = mask_streams["streams"]
            gpt._chain_streams = (lambda ss: (lambda n: ss[:n]))(streams_)      # instance attribute: shadows the probing method
            gpt._mb_shared_queue = False
            if hwid:
                for i in sorted({0, chains - 1}):
                    hwid_report(streams_[i], f"chain {i} ({len(chain_bits[i])} mask bits, {cumode})")
        elif "_chain_streams" in gpt.__dict__:
            del gpt.__dict__["_chain_streams"]      # back to the probed plain streams
            gpt._mb_streams = []
        for k, v in defaults.items():
            L.check(lib.sfmi_tune_set(k.encode(), int(kv.pop(k, v))), f"tune {k}")
        for k, v in kv.items():
            L.check(lib.sfmi_tune_set(k.encode(), int(v)), f"tune {k}")
        gpt._graphs = {}
        if (rows, lclo, lchi) not in cache:
            cache[(rows, lclo, lchi)] = synth_cond(7, rows, lo=lclo, hi=lchi, Lpad=max(406, lchi + 1))
        tok, Lc = cache[(rows, lclo, lchi)]
        ms = []
        try:
            bginfo = ""
            if bg:
                nshape, nl = (int(v) for v in (bg.split(":") + ["60"])[:2])
                bg_setup(nshape)
                bgst["stream"].synchronize()
                t_alone = []
                for _ in range(2):          # the same launch with the chip to itself
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(bgst["stream"])
                    e1 = bg_launch(3)[-1]; e1.synchronize(); t_alone.append(e0.elapsed_time(e1) / 3)
            for rep in range(a.reps + 1):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                bgev = []

                def started():
                    ev[0].record()
                    if bg and rep:
                        bgst["stream"].wait_event(ev[0])
                        bgev.extend(bg_launch(nl))
                kw = dict(max_steps=a.steps, stop_early=False, seed=rep, after_prefill=started)
                if rep:
                    probe.start()
                    if gpt._profile:
                        gpt.launch_profile(reset=True)
                if chains > 1:
                    r = gpt.sample_microbatched(tok, Lc, n_micro=chains, **kw)
                else:
                    r = gpt.sample(tok, Lc, to_host=False, **kw)
                ev[1].record()
                ev[1].synchronize()
                pw = probe.stop() if rep else ""
                done_bg = sum(1 for e in bgev if e.query())
                torch.cuda.synchronize()
                assert int(r["steps"]) == a.steps
                if rep:
                    ms.append(ev[0].elapsed_time(ev[1]) / a.steps)
                    if bg:
                        ar_ms = ev[0].elapsed_time(ev[1])
                        bginfo = (f"   bg SDF {nshape} shapes/launch: alone {min(t_alone):.2f} ms, {done_bg} of {nl} launches done inside the {ar_ms:.0f} ms loop "
                                  f"= {done_bg * min(t_alone):.0f} ms of decode work hidden ({done_bg * nshape / ar_ms * 1e3:.0f} shapes/s of SDF)")
            sem = gpt._sem.cpu().tolist()
            if gpt._profile:
                pr = gpt.launch_profile(reset=True)
                bginfo += "   in situ: " + ", ".join(f"{k} {n} launches x {us:.2f} us" for k, (n, us) in pr.items() if n)
            say(f"{name:28s} rows {rows} chains {chains} {' '.join(kvs):50s} ms/step " + " ".join(f"{m:.3f}" for m in ms)
                + f"   rows/ms {rows / min(ms):.1f}" + (f"   turnstile tickets {sem[0]} time-outs {sem[2]}" if gpt.ATTN_LANES else "") + bginfo + pw)
        except Exception as e:   # keep sweeping
            say(f"{name:28s} FAILED: {type(e).__name__}: {e}")
            torch.cuda.synchronize()
    gpt._ablate, gpt._profile = "", ""


if __name__ == "__main__":
    main()
