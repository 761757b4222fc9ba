#!/usr/bin/env python3
"""Training-step benchmark: ShapeFormer transformer (20+4 layers, d=1024, 325 M params) on
synthetic IMNet-style token batches, per-GPU batch B (YAML: 1), N GPUs via torchrun (one process per GPU, gradients
all-reduced over RCCL as ONE flat 1.3 GB buffer).  Prints one JSON line on rank 0.

    python tools/bench_train.py --batch 1 --steps 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_train.py --batch 1
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def synth_tokens(seed, B, Lc, Lz):
    """(pos,val) rows like the representer emits: ascending positions, end-token padded (representers.py:79-103)."""
    rs = np.random.RandomState(seed)
    def rows(L):
        out = np.full((B, L, 2), 4096, np.int64)
        for b in range(B):
            n = rs.randint(L // 2, L)           # ragged: pad with end tokens like batch_dense2sparse does
            out[b, :n, 0] = np.sort(rs.choice(4096, n, replace=False)); out[b, :n, 1] = rs.randint(0, 4096, n)
        return out
    return torch.from_numpy(rows(Lc)), torch.from_numpy(rows(Lz))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1); ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2); ap.add_argument("--lc", type=int, default=200); ap.add_argument("--lz", type=int, default=300)
    ap.add_argument("--model", default="gpt", choices=["gpt", "vqdif"], help="gpt: CondTupleGPT step; vqdif: VQDIF-16 autoencoder step (SURVEY f4)")
    a = ap.parse_args()
    if a.model == "vqdif":
        return main_vqdif(a)
    from shapeformer_amd import dist as D
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.train import GPTTrainer
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    rank, world, dist = D.init_from_env()
    dev = torch.device("cuda", local)
    g = CondTupleGPT(device=dev)
    tr = GPTTrainer(g, lr=1e-5, dist=dist)
    c, z = synth_tokens(1000 + rank, a.batch, a.lc, a.lz)
    losses = []
    for i in range(a.warmup):
        losses.append(tr.training_step(c, z).item())
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = tr.training_step(c, z)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    dt = time.perf_counter() - t0
    losses.append(loss.item())
    if rank == 0:
        tok = world * a.batch * (a.lc + a.lz - 1) * a.steps
        print(json.dumps({"metric": "training tokens/s (CondTupleGPT 20+4 layers d1024, fwd+bwd+AdamW)", "value": round(tok / dt, 1),
                          "unit": "tokens/s", "n_gpus": world, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 2),
                          "model_TFLOPs": round(6 * 324.95e6 * tok / dt / 1e12, 2), "dtype": "f32", "data": "synthetic",
                          "config": {"batch_per_gpu": a.batch, "L_c": a.lc, "L_z": a.lz, "parallelism": f"dp{world}"},
                          "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4)}))
    if dist:
        dist.destroy_process_group()


def main_vqdif(a):
    """VQDIF-16 training step (vqdif.py:93-137) at the shapenet YAML sizes: Xbd 32768 pts, 8192 query points per shape."""
    from shapeformer_amd import dist as D, synthetic, weights as W
    from shapeformer_amd.train_vqdif import VQDIFTrainer
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    rank, world, dist = D.init_from_env()
    dev = torch.device("cuda", local)
    tr = VQDIFTrainer(W.make_state_dict(W.vqdif_spec(16)), res=16, device=dev, lr=1e-4, beta=0.001, dist=dist)
    b = synthetic.make_batch(2000 + rank * 64, a.batch)
    rs = np.random.RandomState(rank)
    Xtg = rs.uniform(-1, 1, (a.batch, 8192, 3)).astype(np.float32)
    batch = dict(Xbd=torch.from_numpy(b["Xbd"]).to(dev), Xtg=torch.from_numpy(Xtg).to(dev),
                 Ytg=torch.from_numpy((np.linalg.norm(Xtg, axis=-1, keepdims=True) < 0.5).astype(np.float32)).to(dev))
    first = None
    for _ in range(a.warmup):
        out = tr.training_step(batch)
        first = first if first is not None else float(out["loss"])
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.training_step(batch)
    torch.cuda.synchronize()
    if dist: dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "training shapes/s (VQDIF-16 autoencoder, fwd+bwd+Adam+EMA)", "value": round(world * a.batch * a.steps / dt, 2),
                          "unit": "shapes/s", "n_gpus": world, "steps": a.steps, "ms_per_step": round(dt / a.steps * 1e3, 2), "dtype": "f32",
                          "data": "synthetic", "config": {"batch_per_gpu": a.batch, "boundary_N": 32768, "target_N": 8192, "parallelism": f"dp{world}"},
                          "loss_first": round(first, 4), "loss_last": round(float(out["loss"]), 4)}))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
