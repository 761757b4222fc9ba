#!/usr/bin/env python3
"""Package power of a BARE read stream against the decode attention loop.

The AR loop pulls 31.4 GB of f32 K/V per 384-row step.  `tools/ar_sweep.py` measured the attention family alone at
6.15 TB/s and 1291 W of the 1400 W cap; the open question was how much of that is the price of the BYTES (HBM + fabric)
and how much is on-chip work the kernel could shed.  This driver streams the same bytes with the same access pattern
(tools/ubench/stream_power.hip: 1 KiB per wave-load, 16 waves per workgroup, one (row, head) item per workgroup, nontemporal)
and only adds them up, under the same hwmon probe, plus launch-shape / cache-policy variants.

    python tools/stream_power.py --out gpurun_out/r5/stream_power.txt [--secs 3]

Two launch forms per variant: `big` = one grid over all rows x heads x layers of a step (147 456 workgroups);
`layered` = the product's form, 24 layers x 4 chains = 96 launches of 1536 workgroups per step on two streams (two KV
streams in flight, like the two-lane turnstile).
"""
from __future__ import annotations

import argparse
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

from ar_sweep import PowerProbe  # noqa: E402

SRC = os.path.join(ROOT, "tools", "ubench", "stream_power.hip")
SO = os.path.join(ROOT, "tools", "ubench", "libstream_power.so")


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DSFMI_NO_TUNE", "-o", SO, SRC])
    lib = ctypes.CDLL(SO)
    lib.sp_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_void_p]
    lib.sp_launch.restype = ctypes.c_int
    lib.sp_launch_attn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.sp_launch_attn.restype = ctypes.c_int
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r5/stream_power.txt")
    ap.add_argument("--secs", type=float, default=3.0)
    ap.add_argument("--rows", type=int, default=384)
    ap.add_argument("--L", type=int, default=416, help="cached length (mean over the 512 steps of the bench's conditions)")
    ap.add_argument("--variants", default="16040,16041,16042,16043,16020,16080,8080,8081,8082,8040,4160,4161,4162,4080")
    ap.add_argument("--attn-flags", default="", help="comma list of attention-ingredient FLAGS variants (tools/ubench/stream_power.hip:attn_like_kernel), "
                    "run in the layered launch form after the stream variants, e.g. 0,1,3,7,15,31,63")
    ap.add_argument("--build-only", action="store_true", help="compile tools/ubench/stream_power.hip (no GPU needed) and exit")
    a = ap.parse_args()
    if a.build_only:
        build(); return
    lib = build()
    dev = torch.device("cuda:0")
    H, NL, chains = 16, 24, 4
    items_launch = a.rows // chains * H              # one chain's attention launch
    nitems = a.rows * H * NL
    item_floats = 2 * a.L * 64
    step_bytes = nitems * item_floats * 4
    buf = torch.empty(nitems * item_floats, device=dev, dtype=torch.float32)
    for i in range(0, buf.numel(), 1 << 28):          # random payload: zero-filled data draws less power and clocks higher
        buf[i:i + (1 << 28)].normal_()
    out = torch.zeros(nitems, device=dev)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    fo = open(a.out, "a")

    def say(s):
        print(s, flush=True)
        fo.write(s + "\n"); fo.flush()
    probe = PowerProbe()
    say(f"# stream_power: {a.rows} rows x {H} heads x {NL} layers, L = {a.L}: {nitems} items of {item_floats * 4} B = {step_bytes / 1e9:.2f} GB per step; "
        f"probe {probe.bdf} {sorted(probe.files)}")
    streams = [torch.cuda.Stream() for _ in range(2)]

    qkv = torch.randn(4096 * 192, device=dev)

    def step(variant, form):
        if form == "attn":             # the attention's ingredients on top of the stream (variant = FLAGS), the product's launch form
            for li in range(NL * chains):
                s = streams[li % 2]
                rc = lib.sp_launch_attn(variant, buf.data_ptr(), qkv.data_ptr(), out.data_ptr(), li * items_launch, items_launch, a.L, s.cuda_stream)
                assert rc == 0, (variant, rc)
            return
        if form == "big":
            rc = lib.sp_launch(variant, buf.data_ptr(), out.data_ptr(), 0, nitems, a.L, nitems, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (variant, rc)
        else:
            for li in range(NL * chains):
                s = streams[li % 2]
                rc = lib.sp_launch(variant, buf.data_ptr(), out.data_ptr(), li * items_launch, items_launch, a.L, items_launch, s.cuda_stream)
                assert rc == 0, (variant, rc)

    def run(variant, form):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        step(variant, form)
        torch.cuda.synchronize()
        probe.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in streams:
            s.wait_event(e0)
        n, t0 = 0, time.time()
        while time.time() - t0 < a.secs:
            for _ in range(8):
                step(variant, form)
            n += 8
            if n % 64 == 0:                # bound the queue depth
                torch.cuda.synchronize()
        for s in streams:
            cur.wait_stream(s)
        e1.record(); e1.synchronize()
        pw = probe.stop()
        ms = e0.elapsed_time(e1) / n
        if form == "attn":
            names = ["dot+dpp", "lds-scores", "max-barrier", "exp-acc", "prologue", "epilogue"]
            what = "+".join(n for i, n in enumerate(names) if variant >> i & 1) or "two-loop stream"
            say(f"attn flags {variant:2d} {what:60s} ms/step {ms:7.3f}  {step_bytes / ms / 1e9:6.3f} TB/s {pw}")
            return
        w, u, p = variant // 1000, variant % 1000 // 10, variant % 10
        pol = {0: "nt", 1: "default", 2: "buffer sc1", 3: "buffer default"}[p]
        say(f"v{variant:05d} {form:8s} {w:2d} waves U={u:<2d} {pol:14s} ms/step {ms:7.3f}  {step_bytes / ms / 1e9:6.3f} TB/s {pw}")

    vs = [int(v) for v in a.variants.split(",") if v]
    for form in ("big", "layered"):
        for v in vs:
            try:
                run(v, form)
            except Exception as e:
                say(f"v{v} {form} FAILED {type(e).__name__}: {e}")
                torch.cuda.synchronize()
    for f in (int(x) for x in a.attn_flags.split(",") if x):
        try:
            run(f, "attn")
        except Exception as e:
            say(f"attn flags {f} FAILED {type(e).__name__}: {e}")
            torch.cuda.synchronize()
    # idle floor: the same probe with nothing running
    probe.start(); time.sleep(2.0)
    say("idle" + probe.stop())


if __name__ == "__main__":
    main()
