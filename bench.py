#!/usr/bin/env python3
"""bench.py — shapes/sec completed (VQDIF-16 encode + 512-step AR sample + 128^3 SDF extract) on MI355X.

One "step" = one batch of `--batch` synthetic partial clouds taken through the whole hot path
(SURVEY.md §8(d) unit of work): encode 16384-point partial cloud -> (pos,code) tokens -> prefill +
512 KV-cached decode steps of the 20+4-layer d=1024 CondTupleGPT (early exit disabled) -> dense code
grid -> UNet3D + Upsampler -> fused SDF query on the 128^3 lattice -> sigmoid occupancy.
Inputs are resident in HBM before the timed region; weights are hash-generated (no checkpoints ship).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: shapes are independent -> each rank processes its own batch, no data-path collective
("weak" scaling); only the timing barrier / max-over-ranks uses RCCL.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=384, help="shapes per GPU per step (default 384 = 4 interleaved decode chains of 96 rows, one per hardware queue, the most a decode launch holds; 320 = the round-2 workload, 192 = round 1)")
    ap.add_argument("--ar-steps", type=int, default=512)
    ap.add_argument("--decode-res", type=int, default=128)
    ap.add_argument("--points", type=int, default=16384)
    ap.add_argument("--micro", type=int, default=None, help="micro-batches of the AR loop (default: ceil(B/64); 2 for 32..64 rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="keep the stage / AR-loop breakdown, skip the per-kernel roofline timings")
    ap.add_argument("--lanes", type=int, default=None, help="attention turnstile lanes of the interleaved decode chains (default: CondTupleGPT.ATTN_LANES; 0 = off)")
    ap.add_argument("--no-subrecords", action="store_true", help="skip the config2 / config3 / config4 / train sub-records (BASELINE configs 2-5 on this GPU)")
    ap.add_argument("--mode", default="complete", choices=["complete", "train"],
                    help="complete: the shapes/s metric (default); train: DDP training step of the transformer (BASELINE config 5)")
    ap.add_argument("--rank-crcs", action="store_true",
                    help="every rank also runs the fixed-seed pass and the line carries each rank's token CRC pair (rank_token_crcs): the N-rank run "
                         "against N single-process runs of the same inputs (--as-rank), tests/test_ddp_gpu.py")
    ap.add_argument("--as-rank", type=int, default=None, help="TEST ONLY: draw the synthetic inputs rank R of an N-rank run draws (single process)")
    ap.add_argument("--share-device", action="store_true",
                    help="TEST ONLY: all ranks use cuda:0 and rendezvous over gloo (exercises the N-rank path on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true",
                    help="TEST ONLY: initialise the RCCL process group even for one rank (exercises init / barrier / all-reduce of the "
                         "N-rank path on a 1-GPU box; run under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--train-batch", type=int, default=1, help="--mode train: sequences per GPU per step (shapenet_scale.yaml: 1)")
    ap.add_argument("--grad-sync", default="ring", choices=["ring", "rs_ag"],
                    help="--mode train: ring = per-bucket all-reduce, every rank updates everything; rs_ag = per-bucket reduce-scatter, "
                         "AdamW on the rank's 1/N shard, all-gather of the updated parameters (north_star's path)")
    ap.add_argument("--train-gemm", default="sk", choices=["sk", "tile"],
                    help="--mode train: GEMM of the step: sk = work-balanced csrc/sgemm_sk.hip with fused GELU epilogues (default), tile = csrc/sgemm.hip + split-K reduce / GELU launches (rounds 2-4)")
    ap.add_argument("--train-side-stream", type=int, default=1, help="--mode train: 1 = weight-gradient GEMMs / column reductions on a second HIP stream (default), 0 = one stream")
    ap.add_argument("--train-graph", type=int, default=0, help="--mode train: 1 = the step replays one captured hipGraph where it can (one rank, no gradient collectives), 0 = eager (default: as fast on the device)")
    ap.add_argument("--train-fused-opt", type=int, default=1, help="--mode train: 1 = AdamW per bucket inside the backward pass (default), 0 = one AdamW launch after it")
    ap.add_argument("--train-lc", type=int, default=200)
    ap.add_argument("--train-lz", type=int, default=300)
    return ap.parse_args()


def launch_ranks(a):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (one process per GPU,
    rendezvous on 127.0.0.1) and hand over to them.  Under torchrun (WORLD_SIZE set) this is a no-op; a WORLD_SIZE that
    disagrees with --gpus, or fewer visible GPUs than ranks, is an error - never a silent 1-rank run."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={env_world}")
        return
    if a.gpus == 1:
        return
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus and not (a.share_device and n_dev >= 1):
        raise SystemExit(f"bench.py: --gpus {a.gpus} requested but only {n_dev} GPU(s) are visible")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    os.execvpe(sys.executable, cmd, env)


def ev_time(fn, n, warm=2):
    """Average ms per call measured with HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def kernel_rooflines(vq, gpt, B, dev, lc_mean=150.0):
    """Per-kernel live timings (HIP events) at the bench shapes + algorithmic bytes/flops."""
    from shapeformer_amd import ops
    from shapeformer_amd import _lib as L_
    out = []
    D = gpt.D
    st = gpt._alloc(B, 512)
    ly = gpt.layers[0]
    HBM, F32 = 8000.0, 157.3  # GB/s, TFLOP/s peaks (MI355X_MICROARCH.md)

    def add(name, ms, bound, alg, unit_scale, peak, unit, note):
        ach = alg / (ms * 1e-3) / unit_scale
        out.append(dict(kernel=name, bound=bound, ms=round(ms, 5), achieved=round(ach, 2), peak=peak, unit=unit,
                        frac=round(ach / peak, 4), note=note))

    # decode-step weight-streaming GEMMs (dominant by time).  Each is timed as a hipGraph of one launch per
    # transformer layer (24 different weight matrices back-to-back, exactly as in the real step, so the 256 MB
    # Infinity Cache cannot serve them); algorithmic bytes = N*K*4 weights + M*K*4 activations + M*N*4 outputs.
    r = st["resid"]
    for nm, attr, c1a, c2a, xin, res, outb, N, K, ldo, ln, act, S in (
            ("dgemm fc1 (LN+1024->4096+GELU)", "pfc1", "c1fc1", "c2fc1", r, None, st["h"], 4 * D, D, 4 * D, 1, 1, 1),
            ("dgemm fc2 (4096->1024+resid)", "pfc2", None, "bfc2", st["h"], r, r, D, 4 * D, D, 0, 0, gpt.S_FC2),
            ("dgemm qkv (LN+1024->3072)", "pqkv", "c1qkv", "c2qkv", r, None, st["qkv"], 3 * D, D, 3 * D, 1, 0, 1),
            ("dgemm proj (1024->1024+resid)", "pproj", None, "bproj", st["y"], r, r, D, D, D, 0, 0, gpt.S_PROJ if B <= 16 else gpt.S_PROJ_M)):
        def body():
            for l in gpt.layers:
                gpt._dgemm(xin, getattr(l, attr), getattr(l, c1a) if c1a else None, getattr(l, c2a), res, outb, B, N, K, ldo, ln, act, 1, S)
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        ms = ev_time(g.replay, 10) / len(gpt.layers)
        add(nm, ms, "hbm", N * K * 4 + B * K * 4 + B * N * 4, 1e9, HBM, "GB/s", f"M={B}, split-K {S}, {len(gpt.layers)} layers/graph")
    # aggregate of the four launches = the `dgemm_kernel` symbol as rocprofv3 reports it
    gm = out[-4:]
    tot_ms = sum(k["ms"] for k in gm)
    tot_b = sum(k["achieved"] * 1e9 * k["ms"] * 1e-3 for k in gm)
    flops = 2.0 * B * 12.58e6   # 4 weight matrices of one block = 12.58 M parameters
    if 2.0 * B / 4.0 > F32 * 1e3 / HBM:   # arithmetic intensity (FLOP per weight byte) above the machine balance -> MFMA-bound
        add("dgemm_kernel (qkv+proj+fc1+fc2 per layer)", tot_ms, "mfma", flops, 1e12, F32, "TFLOP/s",
            f"M={B}: {2.0 * B / 4.0:.0f} FLOP/B > balance {F32 * 1e3 / HBM:.1f}; weights+activations {tot_b / (tot_ms * 1e-3) / 1e9:.0f} GB/s")
    else:
        add("dgemm_kernel (qkv+proj+fc1+fc2 per layer)", tot_ms, "hbm", tot_b, 1e9, HBM, "GB/s",
            f"M={B}; {tot_b / 1e6:.1f} MB algorithmic per layer; f32 MFMA {flops / (tot_ms * 1e-3) / 1e12:.1f} TFLOP/s")
    out[-1]["launches"] = 4
    # KV-cached decode attention at the mid-run length (Lc + ar_steps/2): bytes = K+V rows of every (row, head)
    saved_len = st["len"].clone()
    st["len"].fill_(int(round(lc_mean)) + 256)      # every row at the mid-run cached length
    st["Kc"].zero_(); st["Vc"].zero_()
    Lavg = float(int(round(lc_mean)) + 256)

    def abody():
        for li in range(len(gpt.layers)):
            L_.check(L_.lib().sfmi_gpt_attn_decode_f32(L_.ptr(st["qkv"]), L_.ptr(gpt.zero_bqkv), L_.ptr(st["Kc"][li]), L_.ptr(st["Vc"][li]),
                                                      L_.ptr(st["len"]), L_.ptr(st["y"]), 1, B, D, gpt.H, gpt.Lmax + 1, None, L_.stream_ptr()), "attn")
    abody(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        abody()
    ms = ev_time(g.replay, 10) / len(gpt.layers)
    st["len"].copy_(saved_len)
    add("attn_decode_kernel", ms, "hbm", B * Lavg * D * 4 * 2, 1e9, HBM, "GB/s", f"{B} rows x {gpt.H} heads, mean cached length {Lavg:.0f}")
    # SDF query (north-star kernel): MFMA-bound, 31 488 FLOP/pt
    Q = 128
    grid = torch.randn(B, 64, 64, 64, 32, device=dev)
    axis = torch.linspace(-1, 1, Q, device=dev)
    o = torch.empty(B, Q ** 3, 1, device=dev)
    ms = ev_time(lambda: ops.sdf_query_grid(axis, grid, vq.sdf_w, sigmoid=True, out=o), 5)
    add("sdf_query_kernel<grid> 128^3", ms, "mfma", B * Q ** 3 * 31488, 1e12, F32, "TFLOP/s",
        f"{B}x128^3 pts; algorithmic HBM {(B * Q ** 3 * 4 + B * 33.55e6) / (ms * 1e-3) / 1e9:.0f} GB/s")
    del grid, o
    # ---- the remaining kernels of the path (north-star names: local-pool encoder, codebook argmin; plus convs, sampler, prefill attention)
    from shapeformer_amd import synthetic
    T = 16384
    X = torch.from_numpy(synthetic.make_batch(99, min(B, 64), n_partial=T)["Xct"]).to(dev)
    Be = X.shape[0]
    lib = L_.lib()
    ws = torch.empty(lib.sfmi_enc_workspace_bytes(Be, T), device=dev, dtype=torch.uint8)
    y0 = torch.empty(Be, 32, 32, 32, 64, device=dev)
    msk = torch.empty(Be, 16, 16, 16, device=dev, dtype=torch.uint8)
    cell = torch.empty(Be, T, device=dev, dtype=torch.int32)
    d0 = vq.down[0]
    ms = ev_time(lambda: L_.check(lib.sfmi_encode_points_down_f32(L_.ptr(X), L_.ptr(vq.enc_w), L_.ptr(d0.w), L_.ptr(y0), L_.ptr(msk), L_.ptr(cell),
                                                                  L_.ptr(ws), Be, T, 16, 1, L_.stream_ptr()), "enc"), 5)
    # algorithmic HBM bytes: 4 pool passes x (T*32*4 read + T*32*4 write + T*4 cell ids) + the mean pass (T*32*4 read)
    # + the 32^3 x 64 output of the fused first Downsampler convolution written once (the product route has no dense 64^3 x 32 grid)
    enc_bytes = Be * (4 * (2 * T * 32 * 4 + T * 4) + T * 32 * 4 + 32 ** 3 * 64 * 4)
    add("enc_block_kernel<0..4> + cell sort + enc_down0_sparse (sfmi_encode_points_down_f32)", ms, "hbm", enc_bytes, 1e9, HBM, "GB/s",
        f"{Be} shapes x {T} points: local-pool scatter_max x4 + scatter_mean + first Downsampler conv, {enc_bytes / Be / 1e6:.1f} MB algorithmic per shape")
    del y0
    lat = torch.randn(Be, 16, 16, 16, 128, device=dev)
    ms = ev_time(lambda: vq.quantize_cl(lat), 10)
    add("vq_argmin_kernel", ms, "mfma", 2.0 * Be * 4096 * 4096 * 128, 1e12, F32, "TFLOP/s", f"{Be} x 4096 cells x 4096 codes x d128 (f32 MFMA + running argmin)")
    code = torch.randn(Be, 16, 16, 16, 128, device=dev)
    ms = ev_time(lambda: vq.decoder_grid_cl(code), 3)
    # UNet3D 31.2 GFLOP + Upsampler with the sub-pixel decomposition of its two up-sampled layers (65.2 -> 34.6 GFLOP) per shape
    conv_flop = Be * (31.2e9 + 34.6e9)
    add("conv3d_igemm_kernel (UNet3D + Upsampler, 16 layers + GroupNorm statistics)", ms, "mfma", conv_flop, 1e12, F32, "TFLOP/s",
        f"{Be} shapes, res16 -> 64^3 x 32 grid; FLOPs as executed (sub-pixel up-sampling: 8/27 of the dense count); round 6: the Upsampler's "
        "GroupNorm statistics come from the convolution epilogues; the apply pass of the last GroupNorm (0.7 ms of this line) moves into the "
        "SDF query for lattices below 128^3 (config2)")
    del lat, code
    # sampler (latency-bound): one tuple element for Bk rows
    lg = torch.randn(B, gpt.Vpad, device=dev) * 3
    st["Lc"].fill_(int(round(lc_mean)))
    st["len"].fill_(int(round(lc_mean)) + 8)      # rows in mid-generation (len > Lc >= 1: the kernel reads token len-1)
    st["seq"].zero_()
    ms = ev_time(lambda: L_.check(lib.sfmi_gpt_sample_f32(L_.ptr(lg), L_.ptr(st["seq"]), L_.ptr(st["len"]), L_.ptr(st["Lc"]), None, None, None,
                                                          None, None, None, None, None, 0, 1, B, gpt.V, gpt.Vpad, gpt.Lmax + 1, 0, 4096, 4096, 100, 0.4, 1.0,
                                                          0, 1, 1, 512, 12345, None, 0, 0, B, 0, L_.stream_ptr()), "sample"), 20)
    st["len"].zero_(); st["Lc"].zero_()
    add("sample_kernel (masker + top-k + top-p + inverse CDF)", ms, "hbm", B * gpt.Vpad * 4, 1e9, HBM, "GB/s", f"{B} rows x 4097 logits; latency-bound by design ({ms * 1e3:.1f} us)")
    # prefill attention on the matrix cores: rows of mean condition length
    P = int(round(lc_mean))
    qkvp = torch.randn(Be * P, 3 * D, device=dev)
    yp = torch.empty(Be * P, D, device=dev)
    nv = torch.full((Be,), P, device=dev, dtype=torch.int32)
    Kc, Vc = st["Kc"][0], st["Vc"][0]
    ms = ev_time(lambda: L_.check(lib.sfmi_gpt_attn_prefill_f32(L_.ptr(qkvp), L_.ptr(Kc), L_.ptr(Vc), L_.ptr(nv), L_.ptr(yp), Be, P, D, gpt.H,
                                                                gpt.Lmax + 1, None, 0.0, 0, L_.stream_ptr()), "attn_prefill"), 10)
    # causal: P(P+1)/2 (query, key) pairs x 2 GEMMs x 2 x 64 flops per head
    add("attn_prefill_mfma_kernel", ms, "mfma", Be * gpt.H * (P * (P + 1) / 2) * 4 * 64, 1e12, F32, "TFLOP/s",
        f"{Be} rows x {gpt.H} heads x {P} positions, causal (useful FLOPs only)")
    return out


def pmc_traffic(kernel, B):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/r0N_pmc_traffic_B<rows>.json, newest round first: FETCH_SIZE
    doubled per the gfx950 correction + WRITE_SIZE); only valid for the configuration it was collected on."""
    f = None
    for r in ("r06", "r05", "r04", "r03", "r02", "r01"):                 # newest collection for this launch shape (tools/pmc_traffic.py)
        c = os.path.join(ROOT, "profiles", f"{r}_pmc_traffic_B{B}.json")
        if os.path.exists(c):
            f = c
            break
    if f is None:
        return None
    key = "dgemm_kernel" if kernel.startswith("dgemm_kernel") else "attn_decode_kernel"
    for k, v in json.load(open(f))["kernels"].items():
        if key in k:
            return v["hbm_bytes_per_launch"]
    return None


def kernel_trace_avg(kernel_prefix):
    """Average launch duration (us) of a kernel in the committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/r0N_bench_kernel_trace.txt, newest round first), or None.  NOT a measurement of this run: the profiler's per-dispatch
    completion signals serialise the decode chains' queues, so its average is a launch that has the chip to itself."""
    f = kernel_trace_file()
    if f is None:
        return None
    f = os.path.join(ROOT, f)
    for ln in open(f):
        if kernel_prefix in ln and not ln.startswith(("#", "{")):
            parts = ln.split()
            try:
                return float(parts[-2])
            except (ValueError, IndexError):
                return None
    return None


def kernel_trace_file():
    for r in ("r06", "r05", "r04"):
        f = os.path.join("profiles", f"{r}_bench_kernel_trace.txt")
        if os.path.exists(os.path.join(ROOT, f)):
            return f
    return None


def pmc_traffic_source(B):
    """Where `roofline.traffic` comes from: it is NOT measured in this run (PMC passes need rocprofv3 around the process)."""
    for r in ("r06", "r05", "r04", "r03", "r02", "r01"):
        f = os.path.join("profiles", f"{r}_pmc_traffic_B{B}.json")
        if os.path.exists(os.path.join(ROOT, f)):
            at = json.load(open(os.path.join(ROOT, f))).get("commit")
            return (f"{f}: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH doubled per the gfx950 correction) of "
                    f"tools/pmc_decode.py at {B} rows per launch, one chain" + (f", taken at commit {at}" if at else ", commit not recorded")
                    + "; not re-measured in this run")
    return None


def effective_cores():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota) — the GPU box exposes 256 logical
    CPUs but caps the container at 16 via cpu.max; oversubscribing torch's pool makes the CPU path 100x slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(points, ar_steps, decode_res):
    """Oracle (CPU restatement pinned to the reference) timed on the host cores, bounded sample (~10-30 s)."""
    from oracle import gpt_oracle as GO, tokens_oracle as TO, vqdif_oracle as VO
    from shapeformer_amd import synthetic, weights as W
    cores = effective_cores()
    torch.set_num_threads(cores)
    sdv = VO.to_torch_sd(W.make_state_dict(W.vqdif_spec(16)))
    X = torch.from_numpy(synthetic.make_batch(314, 1, n_partial=points)["Xct"])
    with torch.no_grad():
        t = time.time(); q, mode, enc = VO.quantize_cloud(sdv, X); t_enc = time.time() - t
        tok, m2 = TO.batch_dense2sparse(q.numpy(), 406, (4096, 4096))
        Lc = tok.shape[1]
        t = time.time(); grid = VO.decoder_grid(sdv, VO.get_code(sdv, q)); t_grid = time.time() - t
        nq = 128 ** 3 // 8  # 1/8 of the 128^3 lattice, scaled up
        pts = torch.from_numpy(VO.make_grid(decode_res))[None, :nq]
        t = time.time(); VO.sdf_query(sdv, grid, pts); t_sdf = (time.time() - t) * (decode_res ** 3 / nq)
        # AR sampling in the reference's formulation (full-prefix recompute, shapeformer.py:86-89): time one step at
        # three prefix lengths and integrate over the 512 steps (a full run is ~minutes/sequence on CPU)
        spec = W.gpt_spec()
        sdg = {k: torch.from_numpy(W.make_tensor(k, s)) for k, s in spec.items()}
        cfg = GO.GPTCfg()
        ts = []
        Ls = [Lc, Lc + ar_steps // 2, min(Lc + ar_steps - 1, 811)]
        for Lp in Ls:
            idx = torch.randint(0, 4096, (1, Lp, 2))
            ex = torch.randint(0, 4096, (1, Lp, 1))
            t = time.time(); GO.forward_logits(sdg, cfg, idx, ex, Lc, idx); ts.append(time.time() - t)
        # trapezoid over steps
        t_ar = (ts[0] + ts[1]) / 2 * (ar_steps / 2) + (ts[1] + ts[2]) / 2 * (ar_steps / 2)
        # the fairer second baseline: the SAME oracle with a KV cache - prefill of the condition once (= the
        # no-cache step at L = Lc timed above) + one-token steps timed at three cached lengths, integrated over the steps
        tk = []
        for Lp in Ls:
            x = torch.randn(1, 1, cfg.n_embd)
            kv = lambda n: [(torch.randn(1, cfg.n_head, Lp, cfg.n_embd // cfg.n_head), torch.randn(1, cfg.n_head, Lp, cfg.n_embd // cfg.n_head))
                            for _ in range(n)]
            c0, c1 = kv(cfg.n_layers[0]), kv(cfg.n_layers[1])
            best = 1e9
            for _ in range(3):
                t = time.time()
                x0, _ = GO.stage(sdg, cfg, 0, x, c0); GO.head(sdg, 0, x0)
                x1, _ = GO.stage(sdg, cfg, 1, x0, c1); GO.head(sdg, 1, x1)
                best = min(best, time.time() - t)
            tk.append(best)
        t_ar_kv = ts[0] + (tk[0] + tk[1]) / 2 * (ar_steps / 2) + (tk[1] + tk[2]) / 2 * (ar_steps / 2)
    total = t_enc + t_ar + t_grid + t_sdf
    total_kv = t_enc + t_ar_kv + t_grid + t_sdf
    kvb = dict(value=round(1.0 / total_kv, 6), unit="shapes/s", cores=cores, kind="port",
               sample=(f"same oracle with a KV cache: prefill {ts[0]:.2f}s + one-token step timed at cached L={Ls} "
                       f"({tk[0] * 1e3:.0f}/{tk[1] * 1e3:.0f}/{tk[2] * 1e3:.0f} ms) integrated over {ar_steps} steps ({t_ar_kv:.1f}s) + the same "
                       f"encode / UNet / SDF legs"))
    return kvb, dict(value=round(1.0 / total, 6), unit="shapes/s", cores=cores, kind="port",
                sample=(f"oracle (torch-CPU fp32 restatement pinned to the reference), 1 shape: encode {t_enc:.2f}s + "
                        f"UNet/upsample {t_grid:.2f}s + 1/8 of the 128^3 SDF query scaled ({t_sdf:.2f}s) + no-KV-cache AR "
                        f"step timed at L={Ls} ({ts[0]:.2f}/{ts[1]:.2f}/{ts[2]:.2f}s) integrated over {ar_steps} steps "
                        f"({t_ar:.0f}s)"))


def config3_record(pipe, gpt, Xct, a):
    """BASELINE config 3 (ShapeFormer AR sampling, 512 tokens, batch 16), measured in the same run as the headline:
    (i) `sample_n` route of VisShapeFormer.compute_batch (shapeformer.py:222-260): 16 sequences of ONE shape, the condition
    prefilled once and its keys / values shared by the 16 rows (gpt.sample(shared_prefix=True)); (ii) plain batch 16: 16
    different shapes through the whole path.  HBM-bound at 16 rows: one weight pass + the KV stream per step."""
    HBM = 8000.0
    enc = pipe.encode_cloud(Xct[:1])
    Lc1 = int(enc["Lc"][0])
    S = 16
    ct = enc["c_tokens"][:1].expand(S, -1, -1).contiguous()
    lt = enc["Lc"][:1].expand(S).contiguous()
    kw = dict(max_steps=a.ar_steps, stop_early=False, to_host=False, best_in_first=False)
    w_one = 4.0 * (sum(l.wqkv.numel() + l.wproj.numel() + l.wfc1.numel() + l.wfc2.numel() for l in gpt.layers) + sum(w.numel() for w in gpt.head_w))
    rec = {}
    for name, shared in (("sample_n16_shared_prefix", True), ("sample_n16_expanded", False)):
        ts = []
        for it in range(2):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            r = gpt.sample(ct, lt, seed=it, shared_prefix=shared, after_prefill=lambda: ev[0].record(), **kw)
            ev[1].record(); torch.cuda.synchronize()
            assert int(r["steps"]) == a.ar_steps
            ts.append(ev[0].elapsed_time(ev[1]))
        ms_step = min(ts) / a.ar_steps
        # per step: one pass over the weights + K and V of every row at its mean cached length (shared rows read the condition once)
        Lgen = (a.ar_steps - 1) / 2.0
        kv = 2 * gpt.D * 4 * len(gpt.layers) * ((Lc1 + S * Lgen) if shared else S * (Lc1 + Lgen))
        rec[name] = {"ms_per_step": round(ms_step, 4), "sequences_per_s": round(S / (min(ts) * 1e-3), 2),
                     "ar_loop_ms": round(min(ts), 1), "L_c": Lc1,
                     "roofline": {"bound": "hbm", "achieved": round((w_one + kv) / ms_step / 1e6, 1), "peak": HBM, "unit": "GB/s",
                                  "frac": round((w_one + kv) / ms_step / 1e6 / HBM, 4),
                                  "note": "algorithmic bytes per step = one weight pass (1.24 GB) + f32 K/V of the rows at the mean cached length"}}
    auto_shared = S * Lc1 >= gpt.SHARED_PREFIX_MIN_ROW_TOKENS
    rec["sample_n16_auto"] = {"takes": "sample_n16_shared_prefix" if auto_shared else "sample_n16_expanded",
                              "rule": f"shared_prefix='auto' (what the sample_n callers pass): shared from rows x L_c >= {gpt.SHARED_PREFIX_MIN_ROW_TOKENS} "
                                      f"(here 16 x {Lc1}); the two forms are bit-identical (tests/test_gpt_gpu.py), crossover measured with tools/bench_shared_prefix.py"}
    X16 = Xct[:16].contiguous()
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = pipe.complete(X16, max_steps=a.ar_steps, decode_res=a.decode_res, seed=CHECK_SEED + it, stop_early=False, sigmoid=True)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        if it == 0:      # BASELINE config 3's own batch: its sampled tokens against the committed checksum, like the headline batch's
            rec["batch16_tokens"] = token_checksums(r, 16, a)
    assert int(r["steps"]) == a.ar_steps
    rec["batch16_whole_path"] = {"shapes_per_s": round(16 / min(ts[1:]), 2), "ms_per_step": round(min(ts[1:]) * 1e3, 1),
                                 "note": "16 different shapes, encode -> 512 AR steps (one 16-row chain) -> UNet + 128^3 SDF query"}
    rec["workload"] = "BASELINE config 3: ShapeFormer 24-layer AR sampling, 512 tokens, batch 16, 1 GPU (d = 1024 as in the shipped YAML)"
    return rec


def vqdif_records(vq16, dev):
    """BASELINE configs 2 and 4, measured in the same run: (2) VQDIF-16 reconstruction of 32 full clouds (32768 points) on the 64^3
    target lattice (vqdif.py:243-269: quantize -> sparse -> dense -> decode_index); (4) VQDIF-32 with a 256^3 query lattice,
    batch 8 = 134 M query points (the decoder stress case).  Whole-call times; the SDF-query share is timed separately and
    priced against its governing roofline (f32 MFMA, 31 488 FLOP per point) with the HBM fraction the north star asks for beside it."""
    from shapeformer_amd import ops, synthetic, weights as W
    from shapeformer_amd.pipeline import ShapeCompletion
    from shapeformer_amd.vqdif import VQDIF
    F32, HBM = 157.3, 8000.0
    rec = {}

    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return min(ts)

    def sdf_line(vq, B, Q):
        grid = torch.randn(B, 64, 64, 64, 32, device=dev)
        axis = torch.linspace(-1, 1, Q, device=dev)
        o = torch.empty(B, Q ** 3, 1, device=dev)
        aff = (torch.rand(B, 32, device=dev) + 0.5, torch.randn(B, 32, device=dev)) if Q < vq.AFFINE_IN_QUERY_BELOW_Q else None   # the instance the product takes at this Q
        ms = ev_time(lambda: ops.sdf_query_grid(axis, grid, vq.sdf_w, sigmoid=False, out=o, affine=aff), 2, warm=1)
        pts = B * Q ** 3
        tf = pts * 31488 / (ms * 1e-3) / 1e12
        hb = (pts * 4 + B * 33.55e6) / (ms * 1e-3) / 1e9
        return {"ms": round(ms, 3), "Gpts_per_s": round(pts / (ms * 1e-3) / 1e9, 2),
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": F32, "unit": "TFLOP/s", "frac": round(tf / F32, 4)},
                "hbm_algorithmic": {"achieved": round(hb, 1), "peak": HBM, "unit": "GB/s", "frac": round(hb / HBM, 4),
                                    "note": "4 B out per lattice point + one read of the 33.5 MB feature grid per shape: the fused kernel is compute-bound (1 968 FLOP/B)"}}
    # config 2
    X2 = torch.from_numpy(synthetic.make_batch(2000, 32)["Xbd"]).to(dev)
    pipe16 = ShapeCompletion(vq16, None)
    t = timed(lambda: pipe16.reconstruct(X2, decode_res=64, max_length=512))
    rec["config2"] = {"workload": "BASELINE config 2: VQDIF res16 reconstruction, batch 32, 32768-point clouds, 64^3 targets",
                      "ms_per_batch": round(t * 1e3, 2), "shapes_per_s": round(32 / t, 1), "sdf_query_64cubed": sdf_line(vq16, 32, 64)}
    del X2
    # config 4
    vq32 = VQDIF(res=32, device=dev)
    X4 = torch.from_numpy(synthetic.make_batch(2100, 8)["Xbd"]).to(dev)
    q, _, _ = vq32.quantize_cloud(X4)
    t_enc = timed(lambda: vq32.quantize_cloud(X4))
    t_dec = timed(lambda: vq32.decode_index(q, grid_Q=256), n=2)
    rec["config4"] = {"workload": "BASELINE config 4: VQDIF res32 + 256^3 SDF query lattice, batch 8 (134 M query points)",
                      "encode_quantize_ms": round(t_enc * 1e3, 2), "decode_index_ms": round(t_dec * 1e3, 1),
                      "shapes_per_s": round(8 / (t_enc + t_dec), 2), "sdf_query_256cubed": sdf_line(vq32, 8, 256)}
    del vq32, X4, q
    torch.cuda.empty_cache()
    return rec


def train_record(gpt, a, batches=(1, 8), steps=5, warm=2):
    """BASELINE config 5 on ONE process, measured in the same run as the headline: the CondTupleGPT training step (forward, backward,
    fused AdamW; no gradient collective with one rank) at the YAML's per-GPU batch 1 and at batch 8."""
    from shapeformer_amd.train import GPTTrainer
    F32 = 157.3
    tr = GPTTrainer(gpt, lr=1e-5, dist=None, graph=True)
    rec = {"workload": "BASELINE config 5 on one rank: CondTupleGPT 20+4 layers d1024, fwd+bwd+AdamW, synthetic tokens L_c 200 + L_z 300", "dtype": "f32",
           "step": "ms_per_step = the eager step (three streams; what GPTTrainer does by default); graph_ms_per_step = the same step replayed as ONE captured "
                   "hipGraph (GPTTrainer(graph=True): tokens, dropout seeds and AdamW bias corrections read from device memory), bit-identical, "
                   "host enqueue 14 -> 2 ms per step but no faster on the device: the step is bound by its kernels, not by the host (DESIGN.md 5.5)"}
    for bs in batches:
        c, z = synth_tokens(1000, bs, a.train_lc, a.train_lz)
        t = {}
        for mode in ("eager", "graph"):
            tr.use_graph = mode == "graph"
            for _ in range(warm + (1 if mode == "graph" else 0)):      # graph: the first step of a shape is eager, the second captures
                loss = tr.training_step(c, z)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                loss = tr.training_step(c, z)
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize(); t[mode] = (time.perf_counter() - t0) / steps
            t[mode + "_host"] = t_host / steps
        dt = t["eager"]
        tok = bs * (a.train_lc + a.train_lz - 1)
        tf = 6 * 324.95e6 * tok / dt / 1e12
        rec[f"batch{bs}"] = {"ms_per_step": round(dt * 1e3, 3), "tokens_per_s": round(tok / dt, 1), "loss": round(float(loss.item()), 4),
                             "host_enqueue_ms_per_step": round(t["eager_host"] * 1e3, 3),
                             "graph_ms_per_step": round(t["graph"] * 1e3, 3), "graph_host_enqueue_ms_per_step": round(t["graph_host"] * 1e3, 3),
                             "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": F32, "unit": "TFLOP/s", "frac": round(tf / F32, 4),
                                          "note": "model FLOPs 6 x 324.95 M parameters x tokens (attention FLOPs not counted)"}}
    del tr
    torch.cuda.empty_cache()
    # the gradient-collective path with ONE rank over RCCL (its own process: `--mode train --force-dist` under torch.distributed.run), both
    # synchronisation modes: what the compute stream waits for collectives when there is nobody to talk to - the baseline the first real
    # multi-GPU run's allreduce_wait_ms / param_allgather_wait_ms are read against.  Eager steps (collectives stay outside a graph).
    import socket
    import subprocess
    one = {}
    for mode in ("ring", "rs_ag"):
        try:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
                   str(port), os.path.abspath(__file__), "--gpus", "1", "--force-dist", "--mode", "train", "--grad-sync", mode, "--steps", "5", "--warmup", "2",
                   "--train-lc", str(a.train_lc), "--train-lz", str(a.train_lz)]
            # its own process group, killed as a whole on a time-out: a hung rank must not outlive the launcher that started it
            pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True,
                                  env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
            try:
                out_, err_ = pr.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                one[mode] = {"error": "timed out after 150 s (process group killed)"}
                continue
            ln = [x for x in out_.splitlines() if x.startswith("{")]
            if pr.returncode != 0 or not ln:
                one[mode] = {"error": (err_ or out_)[-300:]}
                continue
            d = json.loads(ln[-1])
            one[mode] = {k: d.get(k) for k in ("ms_per_step", "allreduce_wait_ms", "param_allgather_wait_ms", "host_enqueue_ms_per_step")}
        except Exception as e:      # a record, never a reason to lose the bench line
            one[mode] = {"error": f"{type(e).__name__}: {e}"[:300]}
    rec["rccl_one_rank_batch1"] = dict(one, note="1 rank, RCCL process group forced on (26 bucket collectives per step run for real, over no links): the "
                                                 "exposed waits of a step when communication is free; eager steps")
    return rec


CHECK_SEED = 1000


def token_checksums(r, B, a, seed=None):
    """CRC-32 of the tokens the AR loop sampled in the fixed-seed pass (row 0 and all rows: int32 (pos, val) pairs of the `ar_steps`
    generated positions), compared with the value committed in tests/golden/bench_token_checksums.json for this exact workload
    (the run is deterministic: counter-hash uniforms, no float atomics, fixed summation orders).  A mismatch means the sampled
    sequences changed - a kernel's rounding, the sampler, or the input selection: it is reported in the line AND the run exits
    non-zero after printing it (main: `token_crc_ok` false anywhere in the line) - the checksum is what ties the timed kernels to the
    parity suite."""
    import zlib
    seed = CHECK_SEED if seed is None else seed
    st = r["state"]
    seq, lc = st["seq"].cpu().numpy(), st["Lc"].cpu().numpy()
    rows = [np.ascontiguousarray(seq[b, lc[b]:lc[b] + a.ar_steps]).astype(np.int32) for b in range(seq.shape[0])]
    crc0 = zlib.crc32(rows[0].tobytes())
    crc_all = 0
    for x in rows:
        crc_all = zlib.crc32(x.tobytes(), crc_all)
    key = f"batch{B}_arsteps{a.ar_steps}_points{a.points}_seed{seed}" + (f"_rank{a.in_rank}" if getattr(a, "in_rank", 0) else "")
    path = os.path.join(ROOT, "tests", "golden", "bench_token_checksums.json")
    want = json.load(open(path)).get(key) if os.path.exists(path) else None
    out = {"token_crc32_row0": crc0, "token_crc32_all_rows": crc_all, "token_crc_key": key,
           "token_crc_committed": want, "token_crc_ok": None if want is None else bool(want == [crc0, crc_all])}
    if want is not None and want != [crc0, crc_all]:
        print(f"bench.py: WARNING sampled tokens differ from the committed checksum for {key}: got {[crc0, crc_all]}, committed {want}",
              file=sys.stderr, flush=True)
    return out


def synth_tokens(seed, B, Lc, Lz):
    """(pos,val) rows like the representer emits: ascending positions, end-token padded (representers.py:79-103)."""
    rs = np.random.RandomState(seed)

    def rows(L):
        out = np.full((B, L, 2), 4096, np.int64)
        for b in range(B):
            n = rs.randint(L // 2, L)           # ragged: pad with end tokens like batch_dense2sparse does
            out[b, :n, 0] = np.sort(rs.choice(4096, n, replace=False))
            out[b, :n, 1] = rs.randint(0, 4096, n)
        return out
    return torch.from_numpy(rows(Lc)), torch.from_numpy(rows(Lz))


def main_train(a, rank, world, dev, dist):
    """BASELINE config 5: data-parallel training step of the (20+4)-layer d=1024 CondTupleGPT (forward, backward, bucketed
    gradient all-reduce over RCCL overlapped with the backward, fused AdamW) on synthetic token batches; one process per
    GPU.  A "step" = one optimizer step on `--train-batch` sequences per GPU of L_c + L_z - 1 tokens."""
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.train import GPTTrainer
    g = CondTupleGPT(device=dev)
    tr = GPTTrainer(g, lr=1e-5, dist=dist, single_rank_collectives=a.force_dist, grad_sync=a.grad_sync, profile_waits=True, gemm=a.train_gemm, side_stream=bool(a.train_side_stream), fused_optimizer=bool(a.train_fused_opt),
                    graph=bool(a.train_graph))
    c, z = synth_tokens(1000 + rank, a.train_batch, a.train_lc, a.train_lz)
    losses = []
    for _ in range(a.warmup):
        losses.append(float(tr.training_step(c, z).item()))
    torch.cuda.synchronize()
    tr.buckets.wait_ms(); tr.buckets.gather_ms()     # drop the warm-up steps' records
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.training_step(c, z)
    t_enq = time.perf_counter() - t0       # the host has ENQUEUED all steps (launches are asynchronous); if this is ~ the total, the step is host-bound
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    losses.append(float(loss.item()))
    wait_ms = tr.buckets.wait_ms()        # mean stall of the compute stream in GradBuckets.finish() over the timed steps
    gather_ms = tr.buckets.gather_ms()    # rs_ag: mean wait for the all-gather of the updated parameters
    if rank == 0:
        tok = world * a.train_batch * (a.train_lc + a.train_lz - 1) * a.steps
        print(json.dumps({
            "metric": "training tokens/s (CondTupleGPT 20+4 layers d1024, fwd+bwd+AdamW, DDP)", "value": round(tok / dt, 1),
            "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ShapeFormer DDP training step, synthetic IMNet-style token batches (BASELINE config 5)",
                       "batch_per_gpu": a.train_batch, "L_c": a.train_lc, "L_z": a.train_lz, "parallelism": f"dp{world}",
                       "grad_sync_mode": a.grad_sync, "gemm": a.train_gemm, "side_stream": bool(a.train_side_stream), "fused_optimizer": bool(a.train_fused_opt),
                       "hipgraph_step": bool(tr._graphs),
                       "grad_sync": ("26 gradient buckets (one per block) all-reduced under the backward pass" if a.grad_sync == "ring" else
                                     "26 gradient buckets reduce-scattered under the backward pass, AdamW on the rank's 1/N shard, updated "
                                     "parameters all-gathered in place through the same flat buffer")},
            "allreduce_wait_ms": None if wait_ms is None else round(wait_ms, 3),
            "allreduce_wait_note": ("time per step the compute stream waits in GradBuckets.finish() for gradient collectives that are still "
                                    "running after the last backward kernel = the EXPOSED communication (ms_per_step - this = compute); "
                                    "null without a process group"),
            "param_allgather_wait_ms": None if gather_ms is None else round(gather_ms, 3),
            "host_enqueue_ms_per_step": round(t_enq / a.steps * 1e3, 3),
            "grad_bytes_per_step": int(tr.flat_grad.numel() * 4),
            "model_TFLOPs": round(6 * 324.95e6 * tok / dt / 1e12, 2),
            "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    launch_ranks(a)
    crc_failed = False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if a.share_device:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local} but only {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")      # --force-dist without torchrun: a one-rank group still needs a rendezvous
        os.environ.setdefault("RANK", str(rank)); os.environ.setdefault("WORLD_SIZE", str(world))
        if a.share_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.mode == "train":
        return main_train(a, rank, world, dev, dist)

    from shapeformer_amd import synthetic
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.pipeline import ShapeCompletion, default_chains
    from shapeformer_amd.vqdif import VQDIF

    vq = VQDIF(res=16, device=dev)
    gpt = CondTupleGPT(device=dev)
    if a.lanes is not None:
        gpt.ATTN_LANES = a.lanes
    pipe = ShapeCompletion(vq, gpt)
    B = a.batch
    # synthetic partial clouds; keep only shapes whose condition length leaves room for ALL ar_steps inside the
    # 812-token block (the unit of work is exactly 512 sampled tuples), decided before the timed region
    in_rank = a.in_rank = rank if a.as_rank is None else a.as_rank      # whose inputs: rank r of any run draws the same shapes
    kept, seed0 = [], 314 + in_rank * 16 * B
    while sum(k.shape[0] for k in kept) < B:          # any rank, any seed: keep drawing until B shapes qualify
        n = max(16, B // 4)
        cand = torch.from_numpy(synthetic.make_batch(seed0, n, n_partial=a.points)["Xct"]).to(dev)
        seed0 += n
        lcs = torch.cat([pipe.encode_cloud(cand[i:i + 64])["Lc"].clone() for i in range(0, n, 64)])
        kept.append(cand[torch.nonzero(lcs <= gpt.Lmax - a.ar_steps).flatten()])
        assert seed0 < 314 + (in_rank + 1) * 16 * B, "synthetic generator yields too few shapes with a short enough condition"
    Xct = torch.cat(kept)[:B].contiguous()
    del kept, cand

    def step(i, timings=None):
        return pipe.complete(Xct, max_steps=a.ar_steps, decode_res=a.decode_res, seed=i, stop_early=False, sigmoid=True, n_micro=a.micro,
                             timings=timings)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # in-situ launch timing of the decode attention (csrc/gpt.hip:prof_begin / prof_end_last): every launch of the TIMED region adds
    # (last workgroup's end - earliest workgroup's start) to a device counter; two atomics per launch on top of the turnstile's
    # finished-workgroup count that runs anyway.  `roofline` below is computed from this average, not from an isolated run.
    gpt._profile = "" if a.no_roofline else "attn"
    for i in range(a.warmup):
        r = step(i)
    barrier()
    gpt.launch_profile(reset=True)
    t0 = time.perf_counter()
    for i in range(a.steps):
        r = step(a.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    prof_timed = gpt.launch_profile(reset=True)
    if dist is not None:
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert int(r["steps"]) == a.ar_steps, f"only {r['steps']} of {a.ar_steps} AR steps were run"
    rank_crcs = None
    if a.rank_crcs:      # every rank's tokens of the fixed-seed pass (no data-path collective: each rank's result is its single-process result)
        tc = token_checksums(step(CHECK_SEED), B, a)
        mine = [tc["token_crc32_row0"], tc["token_crc32_all_rows"]]
        rank_crcs = [mine]
        if dist is not None:
            rank_crcs = [None] * world
            dist.all_gather_object(rank_crcs, mine)
    occ = r["occupancy"]
    sanity = dict(ar_steps_done=int(r["steps"]), occ_mean=round(float(occ.mean().item()), 4),
                  Lc_mean=round(float(r["Lc"].float().mean().item()), 1))

    if rank == 0:
        line = {
            "metric": "shapes/sec completed (64^3 VQDIF + 512-tok AR sample + 128^3 SDF extract)",
            "value": round(world * B * a.steps / dt, 4), "unit": "shapes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"shape completion, {B} shapes/GPU/step: VQDIF-16 encode of {a.points}-pt partial cloud -> "
                                    f"tokens -> CondTupleGPT 20+4 layers d1024 prefill + {a.ar_steps} KV-cached decode steps "
                                    f"(top_k 100, top_p 0.4, early exit off) -> UNet3D+Upsampler -> {a.decode_res}^3 SDF query"),
                       "batch_per_gpu": B, "micro_batches": a.micro or default_chains(B), "ar_steps": a.ar_steps, "decode_res": a.decode_res, "parallelism": f"shard{world}",
                       "weights": "hash-generated (no checkpoints ship)",
                       "input_selection": (f"synthetic partial clouds whose condition length L_c <= {gpt.Lmax - a.ar_steps} (so that all {a.ar_steps} steps fit "
                                           "the 812-token block; biases the cached length down)"),
                       "other_batches": "--batch 320 = the round-2 workload (4 x 80-row chains); --batch 192 = round 1 (4 x 48); --batch 16 = BASELINE config 3's batch (one 16-row chain, also measured in this run: config3)"},
            "sanity": sanity,
        }
        if rank_crcs is not None:
            line["rank_token_crcs"] = rank_crcs
        if not a.no_roofline:
            # one extra, untimed pass with stage marks: where the batch time goes, and the AR loop against its HBM stream
            tm = {}
            gpt._profile = "attn,gemm"       # this extra pass also times the decode-GEMM launches in situ (one more atomic per workgroup)
            step(CHECK_SEED)                 # untimed: re-captures the decode graphs with the instrumented GEMM launches
            gpt.launch_profile(reset=True)
            rd = step(CHECK_SEED, timings=tm)    # fixed sampler seed: the tokens of this pass are checked against a committed checksum
            line["sanity"].update(token_checksums(rd, B, a))
            prof_diag = gpt.launch_profile(reset=True)
            gpt._profile = ""
            lc = r["Lc"].float()
            kv_bytes = float((2 * (lc + (a.ar_steps - 1) / 2.0) * gpt.D * 4 * len(gpt.layers)).sum().item())   # mean over the steps
            n_chain = a.micro or default_chains(B)
            w_one = 4.0 * (sum(l.wqkv.numel() + l.wproj.numel() + l.wfc1.numel() + l.wfc2.numel() for l in gpt.layers)
                           + sum(w.numel() for w in gpt.head_w))
            ms_step = tm["ar_loop"] / a.ar_steps
            line["stages_ms"] = {k: round(v, 1) for k, v in tm.items()}
            if getattr(gpt, "_chain_probe", None):   # stream-pair probe times of gpt._chain_streams (0.2 ms = concurrent)
                line["chain_stream_probe_ms"] = gpt._chain_probe
            sem = gpt._sem.cpu().tolist()            # attention turnstile of the last pass: tickets taken / launches finished / gate time-outs
            line["turnstile"] = {"lanes": gpt.ATTN_LANES, "shared_queue": bool(getattr(gpt, "_mb_shared_queue", False)),
                                 "tickets": sem[0], "finished": sem[1], "timeouts": sem[2]}
            alg, streamed = kv_bytes + w_one, kv_bytes + n_chain * w_one
            line["ar_loop"] = {"ms_per_step": round(ms_step, 3),
                               "algorithmic_bytes_per_step": int(alg), "algorithmic_TBps": round(alg / ms_step / 1e9, 3),
                               "frac_of_hbm_peak": round(alg / ms_step / 1e9 / 8.0, 4),
                               "streamed_bytes_per_step": int(streamed), "streamed_TBps": round(streamed / ms_step / 1e9, 3),
                               "note": ("algorithmic = f32 KV cache of every row at its mean length + ONE pass over the weights; streamed = "
                                        f"the same with one weight pass per decode chain ({n_chain}); HBM peak 8 TB/s (6.3 achievable)")}
            # the loop split into its two kernel families, timed live with the other family disabled (timing-only ablation
            # of gpt.decode_step: results are garbage, launches / bytes / flops are the real ones)
            if not a.no_kernels:
                split = {}
                for fam in ("gemm", "attn"):
                    gpt._ablate = fam          # timing-only ablation (part of the graph cache key)
                    t2 = {}
                    step(a.warmup + a.steps + 1, timings=t2)
                    split[fam] = t2["ar_loop"] / a.ar_steps
                gpt._ablate = ""
                # the attention turnstile (a device-side FIFO gate in front of every attention launch) has to keep earning its place:
                # the same loop with the gate off, in the same run on the same box
                if gpt.ATTN_LANES > 0 and (a.micro or default_chains(B)) > gpt.ATTN_LANES:
                    lanes_on = gpt.ATTN_LANES
                    ab = {}
                    for nm, ln in (("off", 0), (f"lanes{lanes_on}", lanes_on)):
                        gpt.ATTN_LANES = ln
                        step(a.warmup + a.steps + 2)           # re-captures the chains' graphs with / without the gate
                        t3 = {}
                        step(a.warmup + a.steps + 3, timings=t3)
                        ab[nm] = round(t3["ar_loop"] / a.ar_steps, 3)
                    gpt.ATTN_LANES = lanes_on
                    ab["timeouts"] = int(gpt._sem.cpu().tolist()[2])
                    line["turnstile"]["ab_ms_per_step"] = ab
                flops_step = 2.0 * B * (w_one / 4.0)
                line["ar_loop"]["attention_only_ms_per_step"] = round(split["gemm"], 3)
                line["ar_loop"]["attention_only_KV_TBps"] = round(kv_bytes / split["gemm"] / 1e9, 3)
                line["ar_loop"]["gemm_only_ms_per_step"] = round(split["attn"], 3)
                line["ar_loop"]["gemm_only_TFLOPs"] = round(flops_step / split["attn"] / 1e9, 1)
                line["ar_loop"]["split_note"] = ("all chains interleaved, one kernel family disabled at a time (each still runs the head GEMMs + samplers): "
                                                 "the KV stream alone runs at the achievable HBM rate; the two families time-share the chip "
                                                 "(sum ~ real: one attention launch owns all 32 wave slots of every CU, GEMM workgroups of the other chains enter only in its tail - "
                                                 "serialisation by wave-slot exhaustion, not the power cap: profiles/r05_stream_power.md)")
            nm = a.micro or default_chains(B)
            Bk = -(-B // nm)     # rows per decode launch (micro-batch)
        if not a.no_roofline:
            # ---- `roofline`: the dominant kernel of the timed configuration, IN SITU --------------------------------------
            HBM, F32 = 8000.0, 157.3
            n_lay = len(gpt.layers)
            n_a, us_a = prof_timed["attn"]                  # every attention launch of the timed region
            n_g, us_g = prof_diag["gemm"]                   # every decode-GEMM launch of the extra pass (same configuration)
            _, us_a_diag = prof_diag["attn"]
            attn_bytes = kv_bytes / (n_lay * n_chain)       # algorithmic bytes per launch: f32 K+V of the chain's rows at their mean cached length
            gemm_launches_per_step = 4 * n_lay + 2          # qkv, proj, fc1, fc2 per block + the two heads
            gemm_flop = 2.0 * Bk * (w_one / 4.0) / gemm_launches_per_step      # mean FLOP per launch (2 x rows x parameters / launches)
            att = {"kernel": f"attn_decode_kernel ({Bk} rows x {gpt.H} heads per launch)", "bound": "hbm", "launches": n_a,
                   "avg_us": round(us_a, 2), "algorithmic_bytes": int(attn_bytes),
                   "achieved": round(attn_bytes / (us_a * 1e-6) / 1e9, 1) if n_a else None, "peak": HBM, "unit": "GB/s"}
            att["frac"] = round(att["achieved"] / HBM, 4) if n_a else None
            gem = {"kernel": f"dgemm_kernel ({Bk} rows per launch; qkv / proj / fc1 / fc2 / heads)", "bound": "mfma", "launches": n_g,
                   "avg_us": round(us_g, 2), "algorithmic_flop": int(gemm_flop),
                   "achieved": round(gemm_flop / (us_g * 1e-6) / 1e12, 2) if n_g else None, "peak": F32, "unit": "TFLOP/s",
                   "timing": "in situ, extra untimed pass of the same configuration with the GEMM launches instrumented as well"}
            gem["frac"] = round(gem["achieved"] / F32, 4) if n_g else None
            # dominant = the family that takes longer when it runs ALONE in this loop (ar_loop.attention_only / gemm_only of this very
            # run: 5.0 against 2.6 ms per step).  The in-situ launch-time totals of the two families are within 1 % of each other
            # (12.5 ms of launch time per step each, chains overlapping) and flipped the choice from run to run.
            alone = line.get("ar_loop", {})
            if alone.get("attention_only_ms_per_step") and alone.get("gemm_only_ms_per_step"):
                dom_is_attn = n_a > 0 and alone["attention_only_ms_per_step"] >= alone["gemm_only_ms_per_step"]
            else:
                dom_is_attn = n_a > 0 and (n_g == 0 or us_a_diag * prof_diag["attn"][0] >= us_g * n_g)
            dom = att if dom_is_attn else gem
            line["roofline"] = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"],
                                "frac": dom["frac"], "traffic": pmc_traffic(dom["kernel"], Bk), "kernel": dom["kernel"],
                                "traffic_source": pmc_traffic_source(Bk), "rows_per_launch": Bk, "launches": dom["launches"],
                                "avg_us": dom["avg_us"],
                                "timing": ("IN SITU: average over every launch of the timed region, measured by the kernel itself - last workgroup's "
                                           "end minus earliest workgroup's start on the 100 MHz device clock (what a kernel trace reports minus the "
                                           f"dispatch ramp; rocprofv3 --kernel-trace --stats of this command: {kernel_trace_file()})"),
                                "dgemm_in_situ": gem if dom_is_attn else None, "attn_in_situ": None if dom_is_attn else att}
            # how many launches of the dominant kernel are in flight on average (the chains run on separate hardware queues: with the
            # two-lane turnstile two attention launches share the HBM stream), and the rate all of them reach together
            steps_total = a.steps * a.ar_steps
            if dom_is_attn and n_a:
                inflight = n_a * us_a * 1e-3 / (dt * 1e3 * (tm["ar_loop"] / sum(tm.values())))      # launch-time / wall time of the AR loops
                line["roofline"]["mean_launches_in_flight"] = round(inflight, 2)
                line["roofline"]["all_launches_together"] = {
                    "achieved": round(att["achieved"] * inflight, 1), "unit": "GB/s", "frac": round(att["achieved"] * inflight / HBM, 4),
                    "note": "per-launch rate x launches in flight = this kernel's algorithmic bytes over the wall time of the AR loop (GEMM launches of the other chains run in the same interval)"}
            line["roofline"]["summary"] = None       # filled below once the isolated and traced figures are known
            ta = kernel_trace_avg("attn_decode_kernel" if dom_is_attn else "dgemm_kernel")
            if ta:
                alg = attn_bytes if dom_is_attn else gemm_flop
                sc, pk = (1e9, HBM) if dom_is_attn else (1e12, F32)
                line["roofline"]["kernel_trace"] = {
                    "avg_us": ta, "achieved": round(alg / (ta * 1e-6) / sc, 1), "frac": round(alg / (ta * 1e-6) / sc / pk, 4),
                    "source": f"{kernel_trace_file()} (committed; rocprofv3 --kernel-trace --stats of this command)",
                    "note": ("under the profiler every dispatch carries a completion signal and the chains' queues drain one kernel at a time: "
                             "its average is the launch ALONE on the chip (the in-situ average of that profiled run, printed in the same "
                             "file, agrees with it), not the launch as it runs in the timed region")}
        if not a.no_roofline and not a.no_kernels:
            ks = kernel_rooflines(vq, gpt, Bk, dev, lc_mean=sanity["Lc_mean"])
            iso = {k["kernel"].split(" ")[0]: k for k in ks if k["kernel"].startswith(("dgemm_kernel", "attn_decode_kernel"))}
            ik = iso.get("attn_decode_kernel" if dom_is_attn else "dgemm_kernel")
            if ik:
                line["roofline"]["frac_isolated"] = ik["frac"]
                line["roofline"]["isolated_note"] = "one chain alone, hipGraph of 24 consecutive layers, HIP events (kernels[] below)"
            if "gemm_only_TFLOPs" in line.get("ar_loop", {}):
                # all chains in flight, attention launches disabled: launches of different chains overlap, so the chip-level rate of
                # the GEMM phase is higher than one launch's own rate
                line["roofline"]["dgemm_all_chains_in_flight"] = {"achieved": line["ar_loop"]["gemm_only_TFLOPs"], "unit": "TFLOP/s",
                                                                  "frac": round(line["ar_loop"]["gemm_only_TFLOPs"] / 157.3, 4)}
            rf = line["roofline"]
            rf["summary"] = (f"{rf['kernel'].split(' ')[0]}: {rf['frac']:.2f} of peak per launch IN SITU ({rf['avg_us']:.0f} us, "
                             f"{rf.get('mean_launches_in_flight', 1)} launches in flight share the chip"
                             + (f": {rf['all_launches_together']['frac']:.2f} together" if rf.get("all_launches_together") else "") + "); "
                             f"{rf.get('frac_isolated', float('nan')):.2f} for one chain alone"
                             + (f"; {rf['kernel_trace']['frac']:.2f} in the committed rocprofv3 trace ({rf['kernel_trace']['avg_us']:.1f} us: the "
                                "profiler serialises the chains, and the kernel's own clock agrees with it in that profiled run)"
                                if rf.get("kernel_trace") else "") + " - DESIGN.md section 5.2")
            line["kernels"] = ks
        if world == 1 and not a.no_subrecords:
            # BASELINE configs 2-5 in the same driver run (a few seconds each), so that their numbers are not builder-only
            line.update(vqdif_records(vq, dev))          # config2, config4
            line["config3"] = config3_record(pipe, gpt, Xct, a)
            line["train"] = train_record(gpt, a)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline_kv"], line["cpu_baseline"] = cpu_baseline(a.points, a.ar_steps, a.decode_res)
        print(json.dumps(line), flush=True)
        bad = [k for k, v in (("headline", line.get("sanity", {})), ("config3.batch16", line.get("config3", {}).get("batch16_tokens", {})))
               if v.get("token_crc_ok") is False]
        if bad:
            crc_failed = True
            print(f"bench.py: FATAL sampled tokens differ from tests/golden/bench_token_checksums.json ({', '.join(bad)}): the timed kernels "
                  "no longer produce the sequences the parity suite pinned", file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if crc_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
