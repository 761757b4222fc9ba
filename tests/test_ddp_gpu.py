"""GPU: data-parallel training of the REAL trainers with world_size 2 (BASELINE config 5; trainer.py:22,49-56,93 = Lightning
DDP in the reference).  Two worker processes share cuda:0 and rendezvous over gloo (tests/ddp_worker.py); the parent runs the
same two items as ONE batch of 2.  Checks: the bucketed all-reduce is fired from inside the backward pass in completion order;
the averaged gradient of 2 ranks x batch 1 equals the single-process batch-2 gradient; both ranks end with identical
weights; VQDIF: gradient mean and EMA codebook statistics (count, sum) are reduced over ranks -> identical codebooks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def ranks(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ddp") / "w")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return [np.load(out + f".rank{r}.npz") for r in range(2)]


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_gpt_trainer_two_ranks_equal_one_process_batch_two(dev, ranks):
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    from shapeformer_amd.train import GPTTrainer
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=96)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    t = np.load(os.path.join(G, "gpt_tiny.npz"))
    c, z = torch.from_numpy(t["c_idx"]), torch.from_numpy(t["z_idx"])
    tr = GPTTrainer(g, lr=1e-3)
    w0 = np.concatenate([p.detach().cpu().numpy().ravel() for _, p, _ in tr.params[:24]])
    loss = tr.loss_and_grad(c, z)
    ref = tr.flat_grad.detach().cpu().numpy()
    r0, r1 = ranks
    # the collectives were launched by the backward pass itself (not by finish()), block by block in completion order
    assert int(r0["gpt_pending_before_finish"]) == len(tr.buckets.ranges)
    assert list(r0["gpt_order"]) == ["heads", "L2", "L1", "L0", "emb"] == list(r1["gpt_order"])
    assert np.array_equal(r0["gpt_grad"], r1["gpt_grad"])                    # same averaged gradient on both ranks
    assert abs(0.5 * (float(r0["gpt_loss"]) + float(r1["gpt_loss"])) - float(loss.item())) < 1e-5
    # fp32: the batch-2 weight-gradient GEMMs sum 2x the rows in another order than (rank 0) + (rank 1)
    e = _rel(r0["gpt_grad"], ref)
    assert e < 2e-6, e
    # after two optimizer steps both replicas hold the same weights
    assert np.array_equal(r0["gpt_w"], r1["gpt_w"]) and np.isfinite(float(r0["gpt_loss2"]))
    tr.optimizer_step()
    tr.training_step(c, z)
    w = np.concatenate([p.detach().cpu().numpy().ravel() for _, p, _ in tr.params[:24]])
    # Adam divides by sqrt(v): where a gradient is ~0 its rounding noise decides the sign of a full lr-sized step, so the
    # weights are compared through the update they received (direction), not element by element
    d1, d2 = (r0["gpt_w"] - w0).astype(np.float64), (w - w0).astype(np.float64)
    assert float((d1 * d2).sum() / (np.linalg.norm(d1) * np.linalg.norm(d2))) > 0.9999
    assert float(np.abs(r0["gpt_w"] - w).max()) <= 2 * 2 * 1e-3 + 1e-6


def test_gpt_trainer_reduce_scatter_all_gather_equals_ring(ranks):
    """north_star's gradient path (GradBuckets mode "rs_ag": reduce-scatter per bucket under the backward, AdamW on the rank's
    half of every bucket, all-gather of the updated parameters) leaves, after two optimizer steps, the SAME weights bit for bit as
    the ring all-reduce mode on both ranks; each rank updated (and holds moments for) only its own half."""
    r0, r1 = ranks
    for r in (r0, r1):
        assert np.array_equal(r["gpt_w_rsag"], r["gpt_w_all"])
        assert float(r["rsag_shard_frac"]) == 0.5
        assert 0.45 < int(r["rsag_table_chunks"]) / int(r["ring_table_chunks"]) < 0.62      # half of the update work per rank
    assert np.array_equal(r0["gpt_w_rsag"], r1["gpt_w_rsag"])
    # first moments: each rank's rs_ag buffer is non-zero only where it equals the ring mode's, and the two ranks tile the whole
    m0, m1, mr = r0["gpt_m_rsag"], r1["gpt_m_rsag"], r0["gpt_m_ring"]
    own0, own1 = m0 != 0, m1 != 0
    assert not np.any(own0 & own1) and np.array_equal(np.where(own0, m0, m1), np.where(own0 | own1, mr, 0))
    assert np.count_nonzero(own0 | own1) > 0.5 * np.count_nonzero(mr)
    # optimizer_state() is local (this rank's shard and its ranges: no collective); gather_optimizer_state() is the collective that
    # completes it: the same on every rank, equal to the ring mode's moments
    assert np.array_equal(r0["gpt_m_rsag_state"], r1["gpt_m_rsag_state"]) and np.array_equal(r0["gpt_m_rsag_state"], mr)
    assert r0["rsag_local_state_ranges"].shape == r1["rsag_local_state_ranges"].shape and not np.array_equal(r0["rsag_local_state_ranges"], r1["rsag_local_state_ranges"])
    # the parameter all-gathers overlap the NEXT forward: after a step all five buckets (3 blocks, heads, embeddings) were still in
    # flight, and the weights equal the blocking form's (and, above, the ring mode's) bit for bit
    for r in (r0, r1):
        assert int(r["rsag_gathers_in_flight"]) == 5 and int(r["rsag_blocking_in_flight"]) == 0
        assert np.array_equal(r["gpt_w_rsag_blocking"], r["gpt_w_rsag"])


def test_vqdif_trainer_two_ranks_gradient_mean_and_shared_ema_codebook(dev, ranks):
    from shapeformer_amd import weights as W
    from shapeformer_amd.train_vqdif import VQDIFTrainer
    T = np.load(os.path.join(G, "vqdif_train.npz"))
    Xbd = np.concatenate([T["Xbd"], T["Xbd"][:, ::-1] * np.float32(0.9)], 0)
    Xtg = np.concatenate([T["Xtg"], T["Xtg"][:, ::-1]], 0)
    Ytg = np.concatenate([T["Ytg"], T["Ytg"][:, ::-1]], 0)
    vt = VQDIFTrainer(W.make_state_dict(W.vqdif_spec(16)), res=16, device=dev, lr=1e-3, beta=float(T["beta"]))
    o = vt.training_step(dict(Xbd=Xbd, Xtg=Xtg, Ytg=Ytg))
    r0, r1 = ranks
    assert np.array_equal(r0["vq_grad"], r1["vq_grad"])
    # EMA statistics are all-reduced: every rank keeps the SAME codebook, equal to the one-process batch-2 update
    assert np.array_equal(r0["vq_emb"], r1["vq_emb"]) and np.array_equal(r0["vq_N"], r1["vq_N"])
    assert np.abs(r0["vq_N"] - vt.N.cpu().numpy()).max() < 1e-5
    assert _rel(r0["vq_emb"], vt.emb.cpu().numpy()) < 1e-5
    assert abs(0.5 * (float(r0["vq_loss"]) + float(r1["vq_loss"])) - float(o["loss"])) < 1e-5
    e = _rel(r0["vq_grad"], vt.flat_g.cpu().numpy())
    assert e < 2e-5, e
    assert np.array_equal(r0["vq_p"], r1["vq_p"])
    assert float(np.abs(r0["vq_p"] - vt.flat_p.cpu().numpy()).max()) <= 2 * 1e-3 + 1e-6      # one Adam step of lr 1e-3 (see above)


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` without a torchrun environment must start 2 ranks and report n_gpus 2 (on this 1-GPU box the
    ranks share cuda:0 over gloo: --share-device).  Tiny workload; the value is not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--steps", "1", "--warmup", "0",
                        "--batch", "4", "--ar-steps", "4", "--decode-res", "32", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and line[0]["n_gpus"] == 2 and line[0]["config"]["parallelism"] == "shard2"
    # the training line (BASELINE config 5) the same way: 2 ranks, gradient buckets all-reduced under the backward (gloo here)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--mode", "train", "--steps", "2",
                        "--warmup", "1", "--train-lc", "24", "--train-lz", "40"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and line[0]["n_gpus"] == 2 and line[0]["config"]["parallelism"] == "dp2" and line[0]["unit"] == "tokens/s"
    assert line[0]["allreduce_wait_ms"] is not None and line[0]["allreduce_wait_ms"] >= 0 and line[0]["grad_bytes_per_step"] > 1.2e9
    # north_star's gradient path through the same launcher
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--mode", "train", "--steps", "2",
                        "--warmup", "1", "--train-lc", "24", "--train-lz", "40", "--grad-sync", "rs_ag"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and line[0]["config"]["grad_sync_mode"] == "rs_ag" and line[0]["param_allgather_wait_ms"] is not None
    # and the guard: asking for more ranks than GPUs is an error, never a silent 1-rank run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only" in (r.stdout + r.stderr)


def test_bench_rccl_path_with_one_rank(dev):
    """The RCCL calls of the N-rank bench path (init_process_group("nccl", device_id), barrier, all-reduce MAX of the step time)
    on this 1-GPU box: one rank under torch.distributed.run with --force-dist.  Tiny workload; the value is not a measurement."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "1",
                        "--warmup", "0", "--batch", "4", "--ar-steps", "4", "--decode-res", "32", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and line[0]["n_gpus"] == 1 and line[0]["value"] > 0
    # --mode train under the same launcher: the 26 gradient buckets go through RCCL all_reduce(ReduceOp.AVG, async_op=True) fired
    # from inside the backward (a one-rank group: numerically a no-op, the calls / stream ordering / waits are the real ones)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--mode", "train",
                        "--steps", "2", "--warmup", "1", "--train-lc", "24", "--train-lz", "40"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and line[0]["n_gpus"] == 1 and line[0]["unit"] == "tokens/s" and np.isfinite(line[0]["loss_last"])
    assert line[0]["allreduce_wait_ms"] is not None, "the RCCL bucket path did not run"
    # north_star's gradient path on RCCL: in-place reduce_scatter_tensor / all_gather_into_tensor per bucket (one-rank group: the
    # collectives, their stream ordering and the sharded AdamW + unflatten launches are the real ones)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--mode", "train",
                        "--steps", "2", "--warmup", "1", "--train-lc", "24", "--train-lz", "40", "--grad-sync", "rs_ag"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    l2 = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(l2) == 1 and l2[0]["config"]["grad_sync_mode"] == "rs_ag" and l2[0]["param_allgather_wait_ms"] is not None
    # same data, same seeds, one rank: the sharded update (shard = everything) must reproduce the ring mode's loss trajectory
    assert abs(l2[0]["loss_last"] - line[0]["loss_last"]) < 1e-6 and abs(l2[0]["loss_first"] - line[0]["loss_first"]) < 1e-6


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL across devices)")
def test_rs_ag_equals_ring_over_rccl_on_two_gpus(tmp_path):
    """the in-place reduce_scatter_tensor / all_gather_into_tensor of GradBuckets mode "rs_ag" over REAL RCCL with world size 2
    (the 2-rank tests above run over gloo on one device; the 1-rank RCCL legs cannot see aliasing bugs): two processes, one GPU each,
    run `bench.py --mode train` for a few steps in both modes; the final losses must agree to fp32 rounding (same weights)."""
    import json
    outs = {}
    for mode in ("ring", "rs_ag"):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "1", "--grad-sync", mode]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert abs(outs["ring"]["loss_last"] - outs["rs_ag"]["loss_last"]) < 1e-4 and outs["rs_ag"]["n_gpus"] == 2


def test_sample_n_of_one_shape_split_over_two_ranks(dev, ranks):
    """SURVEY section 8(e), single-shape option (dist.sample_n_sharded): 7 sequences of ONE condition, 3 on rank 0 and 4 on rank 1 with
    their global row indices and a collective early stop, gathered on both ranks == the 7 rows sampled by ONE process, bit for
    bit (tokens, log-probabilities, number of steps), with the early stop on and with all steps forced."""
    from shapeformer_amd import weights as W
    from shapeformer_amd.gpt import CondTupleGPT
    kw = dict(n_embd=128, n_layers=(2, 1), block_size=96)
    g = CondTupleGPT(W.make_state_dict(W.gpt_spec(**kw)), n_embd=128, n_head=2, n_layers=(2, 1), block_size=96, device=dev)
    tk = np.load(os.path.join(G, "vqdif16_small.npz"))["tokens"].astype(np.int64)[0, :23]       # 23 real tokens + the end pair
    c1 = torch.from_numpy(np.concatenate([tk, np.full((1, 2), 4096, np.int64)])[None]).to(torch.int32)
    c7, L7 = c1.expand(7, -1, -1).contiguous(), torch.full((7,), c1.shape[1], dtype=torch.int32)
    for name, skw in (("sn", dict(max_steps=40, seed=9, check_every=4)), ("sn_full", dict(max_steps=24, seed=9, stop_early=False, mask_invalid=False))):
        one = g.sample(c7, L7, **skw)
        for r in ranks:
            assert int(r[name + "_steps"]) == int(one["steps"]), (name, int(r[name + "_steps"]), int(one["steps"]))
            assert np.array_equal(r[name + "_samples"], one["samples"].numpy()), name
            assert np.array_equal(r[name + "_logp"], one["log_prob"].numpy()), name
    assert len(set(map(tuple, one["samples"][1:, :, 0].tolist()))) > 1       # the stochastic rows differ from each other


def test_sdf_lattice_of_one_shape_split_into_slabs_over_two_ranks(dev, ranks):
    """SURVEY section 8(e), the other single-shape option (dist.sdf_query_sharded): the 33^3 occupancy lattice of ONE shape evaluated as two
    slabs of lattice planes (16 on rank 0, 17 on rank 1: uneven on purpose), feature grid replicated, slabs all-gathered == `decode_index`
    of one process bit for bit, logits and fused sigmoid; and the slab entry alone reproduces any plane range of the whole-lattice call."""
    from shapeformer_amd.vqdif import VQDIF
    vq = VQDIF(res=16, device=dev)
    code = torch.from_numpy(np.load(os.path.join(G, "vqdif16_small.npz"))["quant_ind"][:1].astype(np.int64)).to(dev)
    for sg, key in ((False, "sdf_slabs"), (True, "sdf_slabs_sig")):
        one = vq.decode_index(code, grid_Q=33, sigmoid=sg)["logits"].cpu().numpy()
        assert one.shape == (1, 33 ** 3, 1) and np.isfinite(one).all()
        for r in ranks:
            assert np.array_equal(r[key], one), key
    whole = vq.decode_index(code, grid_Q=20)["logits"]
    for x0, x1 in ((0, 1), (3, 11), (19, 20), (0, 20)):
        part = vq.decode_index(code, grid_Q=20, x_range=(x0, x1))["logits"]
        assert torch.equal(part, whole[:, x0 * 400:x1 * 400])


def test_bench_two_ranks_sharing_the_device_sample_what_single_processes_sample(dev):
    """bench.py end to end as the driver launches it for N = 2 (torch.distributed.run, one process per rank; here both ranks on cuda:0
    over gloo): the shapes are sharded with no data-path collective, so each rank's tokens of the fixed-seed pass must be exactly the
    tokens ONE process samples for that rank's inputs (`--as-rank`), and the line's aggregate is over both ranks."""
    import json
    base = ["--batch", "24", "--ar-steps", "48", "--decode-res", "32", "--points", "4096", "--steps", "1", "--warmup", "1", "--no-roofline",
            "--no-subrecords", "--no-cpu-baseline", "--rank-crcs"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + base, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    two = run(["--gpus", "2", "--share-device"])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and len(two["rank_token_crcs"]) == 2
    singles = [run(["--gpus", "1", "--as-rank", str(r)])["rank_token_crcs"][0] for r in range(2)]
    assert two["rank_token_crcs"] == singles, (two["rank_token_crcs"], singles)
    assert singles[0] != singles[1]                      # the ranks really work on different shapes
    assert two["sanity"]["ar_steps_done"] == 48
