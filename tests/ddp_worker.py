"""Worker of tests/test_ddp_gpu.py: one data-parallel rank of the REAL trainers (GPTTrainer / VQDIFTrainer with dist=...).

Launched as `python tests/ddp_worker.py <out.npz>` with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment
(what torchrun sets).  All ranks share cuda:0 and rendezvous over gloo - the collective path taken is the one RCCL takes on
an 8-GPU node (dist.GradBuckets fired from the real backward order, flat-buffer all-reduce, EMA statistics all-reduce)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def main():
    import torch.distributed as dist
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.train import GPTTrainer
    from shapeformer_amd.train_vqdif import VQDIFTrainer
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    out = {}
    # ---- transformer: rank r trains on item r of the 2-item reference token fixture ------------------------------
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=96)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    t = np.load(os.path.join(G, "gpt_tiny.npz"))
    c, z = torch.from_numpy(t["c_idx"])[rank::world], torch.from_numpy(t["z_idx"])[rank::world]
    tr = GPTTrainer(g, lr=1e-3, dist=dist)
    loss = tr.loss_and_grad(c, z, sync=True)          # buckets are launched from inside the backward pass
    out["gpt_pending_before_finish"] = np.int64(len(tr.buckets.pending))
    out["gpt_order"] = np.array(tr.all_reduce_grads())
    out["gpt_loss"] = np.float64(loss.item())
    out["gpt_grad"] = tr.flat_grad.detach().cpu().numpy().copy()
    tr.optimizer_step()
    loss2 = tr.training_step(c, z)                     # second full step through the public entry point
    out["gpt_loss2"] = np.float64(loss2.item())
    out["gpt_w"] = np.concatenate([p.detach().cpu().numpy().ravel() for _, p, _ in tr.params[:24]])
    out["gpt_w_all"] = np.concatenate([p.detach().cpu().numpy().ravel() for _, p, _ in tr.params])
    # ---- the same two steps with north_star's gradient path: reduce-scatter per bucket, AdamW on this rank's shard, all-gather
    # of the updated parameters (GradBuckets mode "rs_ag") - must leave bit-identical weights
    g2 = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    tr2 = GPTTrainer(g2, lr=1e-3, dist=dist, grad_sync="rs_ag")
    assert all(tr2.buckets.sharded(n) for n in tr2.buckets.ranges), "every bucket of this model divides by 2"
    tr2.training_step(c, z)
    slo, shi = tr2.buckets.shard("L1")
    lo, hi = tr2.buckets.ranges["L1"]
    out["rsag_shard_frac"] = np.float64((shi - slo) / (hi - lo))
    # update work per rank: chunks of the per-bucket AdamW tables (fused step), sharded mode against the ring trainer's whole buckets
    out["rsag_table_chunks"] = np.int64(sum(tb["n"] for tb in tr2._opt_tabs.values()))
    out["ring_table_chunks"] = np.int64(sum(tb["n"] for tb in tr._opt_tabs.values()))
    tr2.training_step(c, z)
    # the parameter all-gathers of the step were launched, not waited for: the next forward (or finish_param_gather) consumes them
    out["rsag_gathers_in_flight"] = np.int64(len(tr2.buckets.params_in_flight()))
    tr2.finish_param_gather()
    out["gpt_w_rsag"] = np.concatenate([p.detach().cpu().numpy().ravel() for _, p, _ in tr2.params])
    out["gpt_m_rsag"] = tr2.flat_m.detach().cpu().numpy().copy()     # moments exist only on the rank's shard
    local = tr2.optimizer_state()                                      # local, no collective: this rank's shard + its ranges
    out["rsag_local_state_ranges"] = np.array(local["shard_ranges"], np.int64)
    out["gpt_m_rsag_state"] = tr2.gather_optimizer_state()["exp_avg"].numpy().copy()     # the explicit collective completes it
    # the blocking form (all gathers waited for right after the optimizer) leaves the same weights
    g3 = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    tr3 = GPTTrainer(g3, lr=1e-3, dist=dist, grad_sync="rs_ag", overlap_param_gather=False)
    tr3.training_step(c, z); tr3.training_step(c, z)
    out["rsag_blocking_in_flight"] = np.int64(len(tr3.buckets.params_in_flight()))
    out["gpt_w_rsag_blocking"] = np.concatenate([p.detach().cpu().numpy().ravel() for _, p, _ in tr3.params])
    out["gpt_m_ring"] = tr.flat_m.detach().cpu().numpy().copy()
    # ---- SURVEY 8(e), single-shape option: the sample_n = 7 sequences of ONE condition split over the ranks (3 + 4 rows), early stop on
    from shapeformer_amd.dist import sample_n_sharded
    g4 = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    tk = np.load(os.path.join(G, "vqdif16_small.npz"))["tokens"].astype(np.int64)[0, :23]       # 23 real tokens + the end pair
    c1 = torch.from_numpy(np.concatenate([tk, np.full((1, 2), 4096, np.int64)])[None]).to(torch.int32)
    for name, skw in (("sn", dict(max_steps=40, seed=9, check_every=4)), ("sn_full", dict(max_steps=24, seed=9, stop_early=False, mask_invalid=False))):
        r = sample_n_sharded(g4, c1, torch.tensor([c1.shape[1]], dtype=torch.int32), 7, dist, **skw)
        out[name + "_samples"], out[name + "_logp"], out[name + "_steps"] = r["samples"].numpy(), r["log_prob"].numpy(), np.int64(r["steps"])
    # ---- SURVEY 8(e), the other single-shape option: the 33^3 occupancy lattice of ONE shape in two slabs of planes (16 + 17), gathered
    from shapeformer_amd.dist import sdf_query_sharded
    from shapeformer_amd.vqdif import VQDIF
    vq = VQDIF(res=16, device=dev)
    code = torch.from_numpy(np.load(os.path.join(G, "vqdif16_small.npz"))["quant_ind"][:1].astype(np.int64)).to(dev)
    for sg in (False, True):
        out["sdf_slabs" + ("_sig" if sg else "")] = sdf_query_sharded(vq, code, 33, dist, sigmoid=sg)["logits"].cpu().numpy()
    # ---- VQDIF autoencoder: item r of a 2-item batch; gradients averaged, EMA statistics summed over ranks ------
    T = np.load(os.path.join(G, "vqdif_train.npz"))
    Xbd = np.concatenate([T["Xbd"], T["Xbd"][:, ::-1] * np.float32(0.9)], 0)      # two different clouds
    Xtg = np.concatenate([T["Xtg"], T["Xtg"][:, ::-1]], 0)
    Ytg = np.concatenate([T["Ytg"], T["Ytg"][:, ::-1]], 0)
    vt = VQDIFTrainer(W.make_state_dict(W.vqdif_spec(16)), res=16, device=dev, lr=1e-3, beta=float(T["beta"]), dist=dist)
    o = vt.training_step(dict(Xbd=Xbd[rank::world], Xtg=Xtg[rank::world], Ytg=Ytg[rank::world]))
    out["vq_loss"] = np.float64(float(o["loss"]))
    out["vq_grad"] = vt.flat_g.detach().cpu().numpy().copy()      # after the in-place mean all-reduce of optimizer_step
    out["vq_emb"], out["vq_N"] = vt.emb.detach().cpu().numpy().copy(), vt.N.detach().cpu().numpy().copy()
    out["vq_p"] = vt.flat_p.detach().cpu().numpy().copy()
    np.savez(sys.argv[1] + f".rank{rank}.npz", **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
