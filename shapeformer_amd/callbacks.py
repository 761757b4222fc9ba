"""Inference drivers and result formats around the hot path.

The reference runs inference through Lightning callbacks: `VisSparseRecon3D` (vqdif.py:217-310) and `VisShapeFormer`
(shapeformer.py:209-329) on top of `plutil.VisCallback.process` (xgutils/plutil.py:163-220), which pulls items from a
dataset, calls `compute_batch` (cached as `computed/<name>.npy`), then `visualize_batch`, which decodes every sample to
a 128^3 occupancy grid, extracts a mesh on the CPU (PyMCubes), renders images and writes `meshes/<name>_<key>.ply` and
`eval/<name>.npz` (10^5 surface samples per mesh).

Here the same classes (same ctor kwargs, same `compute_batch` dict keys, same files) run on the MI355X path:
sampling, decoding and iso-surface extraction stay on the device; only tokens, log-probabilities and meshes reach the
host.  Rendering (fresnel) is out of scope: `visualize_batch` returns the meshes instead of images.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import meshio, ops, tokens as T
from .dist import effective_indices


def filter_end_tokens(indices, end_tokens):
    """models/common.py:50-55."""
    indices = np.asarray(indices)
    valid = (indices != np.array(end_tokens)[None, :]).all(axis=1)
    return indices[valid]


def compute_log_probs(samples, logits_history):
    """shapeformer.py:407-418: log-softmax of the recorded (masked) step logits at the sampled tokens -> (S, L, tuple_n)."""
    samples = np.asarray(samples)
    S, Ls, tn = samples.shape
    out = np.zeros(samples.shape)
    for ti in range(tn):
        lg = np.asarray(logits_history[ti], np.float64)[:, :Ls]
        m = lg.max(-1, keepdims=True)
        lse = m + np.log(np.exp(lg - m).sum(-1, keepdims=True))
        out[..., ti] = np.take_along_axis(lg - lse, samples[..., ti][..., None], -1)[..., 0]
    return out


class VisCallback:
    """plutil.VisCallback (xgutils/plutil.py:147-222) minus Lightning hooks and image summaries."""

    def __init__(self, visual_indices=(0, 1, 2, 3, 4, 5), all_indices=False, force_visual_indices=False, every_n_epoch=3,
                 no_sanity_check=False, load_compute=False, load_visual=False, data_dir=None, output_name=None,
                 use_dloader=False, num_gpus=1, parallel_vis=False, single_vis=True, visall_after_training_end=True, **_):
        # plutil.py:177: the YAML value "all" (configs/demo/*.yaml) selects every item, like all_indices
        self.visual_indices = "all" if ((all_indices and not force_visual_indices) or (isinstance(visual_indices, str) and visual_indices == "all")) \
            else list(visual_indices)
        self.force_visual_indices, self.load_compute, self.num_gpus, self.parallel_vis = force_visual_indices, load_compute, num_gpus, parallel_vis
        self.output_name = output_name or type(self).__name__
        self.data_dir = data_dir or os.path.join("experiments", "temp", self.output_name)
        self.pl_module = None

    def process(self, pl_module, dataset, visual_indices=None, data_dir=None, load_compute=None, parallel_vis=None, rank=0):
        """For every selected item: compute (or reload `computed/<ind>.npy`), save it, export meshes / eval samples.
        `dataset[i]` is a dict of numpy arrays (Xct, Xbd, ...) like the reference datasets' `__getitem__`.
        parallel_vis: this rank handles items rank, rank+G, ... (plutil.py:123-139).  Returns {name: exported}."""
        self.pl_module = pl_module
        data_dir = data_dir or self.data_dir
        vi = self.visual_indices if visual_indices is None else visual_indices
        if vi == "all":
            vi = list(range(len(dataset)))
        if (self.parallel_vis if parallel_vis is None else parallel_vis):
            vi = effective_indices(vi, rank, self.num_gpus).tolist()
        load_compute = self.load_compute if load_compute is None else load_compute
        cdir = os.path.join(data_dir, "computed")
        os.makedirs(cdir, exist_ok=True)
        out, failed = {}, []
        for ind in vi:
            name = str(ind)
            try:
                path = os.path.join(cdir, name + ".npy")
                computed = np.load(path, allow_pickle=True).item() if (load_compute and os.path.exists(path)) else None
                if computed is None:
                    item = dataset[ind]
                    batch = {k: torch.from_numpy(np.asarray(v))[None] for k, v in item.items() if isinstance(v, (np.ndarray, torch.Tensor))}
                    computed = self.compute_batch(batch, input_name=name)
                np.save(path, computed)                                   # FlyObj.save (plutil.py:68-72)
                out[name] = self.visualize_batch(computed, input_name=name, data_dir=data_dir)
            except Exception as e:  # the reference logs and continues (plutil.py:199-205)
                import traceback
                traceback.print_exc()
                failed.append(ind)
        os.makedirs(os.path.join(data_dir, "logs", "failed_ind"), exist_ok=True)
        np.savetxt(os.path.join(data_dir, "logs", "failed_ind", f"rank_{rank}.txt"), np.array(failed))
        return out


def _np(d):
    """ptutil.ths2nps."""
    if isinstance(d, dict):
        return {k: _np(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return type(d)(_np(v) for v in d)
    return d.detach().cpu().numpy() if isinstance(d, torch.Tensor) else d


class VisSparseRecon3D(VisCallback):
    """vqdif.py:217-310: quantize -> sparse tokens -> dense -> decode on the `decoder_resolution`^3 lattice -> mesh."""

    def __init__(self, samples=32, Xct_as_Xbd=False, quant_grid_depth=4, decoder_resolution=128, vocab_size=4096,
                 max_length=512, end_tokens=(4096, 4096), resolution=(512, 512), vis_Ytg=True, thresh=0.5, **kw):
        super().__init__(**kw)
        self.Xct_as_Xbd, self.quant_grid_depth, self.decoder_resolution = Xct_as_Xbd, quant_grid_depth, decoder_resolution
        self.vocab_size, self.max_length, self.end_tokens, self.thresh = vocab_size, max_length, tuple(end_tokens), thresh

    @torch.no_grad()
    def compute_batch(self, batch, input_name=""):
        vq = getattr(self.pl_module, "core", self.pl_module)
        Xbd = batch["Xbd"] if ("Xbd" in batch and not self.Xct_as_Xbd) else batch["Xct"]
        q, mode, raw, mask, _ = vq.quantize_cloud_dev(Xbd.to(vq.dev, torch.float32), per_shape_mode=False)
        mode2 = T.mode_i32(q, vq.K + 1)                                       # batch_dense2sparse's own mode (common.py:155)
        tok, ln = T.dense2sparse_dev(q, mode2, self.max_length, self.end_tokens, Lpad=self.max_length)
        dense = T.sparse2dense_dev(tok, ln, mode2, 2 ** self.quant_grid_depth, self.end_tokens)
        logits = vq.decode_index(dense, grid_Q=self.decoder_resolution)["logits"]
        # pack_sparse (common.py:126-140): (K,3) [b, pos, val] without the end-token rows
        th, lh = tok.cpu().numpy().astype(np.int64), ln.cpu().numpy()
        rows = []
        for b in range(th.shape[0]):
            t = filter_end_tokens(th[b, :lh[b]], self.end_tokens)
            rows.append(np.concatenate([np.full((len(t), 1), b, np.int64), t], 1))
        self._occ_dev = ops.sigmoid(logits)[..., 0]                          # stays in HBM for the mesh extraction
        return _np({"logits": logits, "quant_ind": raw.long(), "sparse": np.concatenate(rows, 0),
                    "grid_mask": mask.bool(), "batch": batch})

    def visualize_batch(self, computed, input_name="", data_dir=None):
        from . import mcubes
        data_dir = data_dir or self.data_dir
        Q = self.decoder_resolution
        occ = getattr(self, "_occ_dev", None)
        if occ is None or occ.shape[-1] != Q ** 3:
            dev = getattr(self.pl_module, "core", self.pl_module).dev
            occ = ops.sigmoid(torch.as_tensor(computed["logits"]).to(dev))[..., 0]       # nputil.sigmoid(logits)
        self._occ_dev = None
        v, f, voff, toff = mcubes.marching_cubes_dev(occ[:1].reshape(1, Q, Q, Q), self.thresh)
        vert, face = v.cpu().numpy().astype(np.float64), f.cpu().numpy().astype(int)
        path = meshio.write_mesh(data_dir, vert, face, input_name)                         # geoutil.write_mesh
        out = {"recon_mesh": {"vert": vert, "face": face}, "mesh_path": path}
        if len(face):
            eval_pc = meshio.sample_mesh(vert, face, 10 ** 5)
            os.makedirs(os.path.join(data_dir, "eval"), exist_ok=True)
            np.savez(os.path.join(data_dir, "eval", f"{input_name}.npz"), eval_pc=eval_pc)
            out["eval_pc"] = eval_pc
        return out


class VisShapeFormer(VisCallback):
    """shapeformer.py:209-329: `sample_n` completions of ONE partial cloud, sorted by sequence probability, each decoded
    to a `decode_res`^3 occupancy grid and meshed."""

    def __init__(self, temperature=1, sample_n=10, top_k=300, top_p=.9, depth=5, decode_res=128, sample_max_step=512,
                 render_samples=64, end_tokens=None, mask_invalid=True, mask_invalid_completion=False,
                 force_keep_c_indices=False, sort_prob=True, partial_radius=0.02, camPos=(2, 2, 2), resolution=(512, 512),
                 thresh=0.5, keep_logits_history=False, seed=0, shard_sample_n=None, **kw):
        super().__init__(**kw)
        # shard_sample_n: a torch.distributed module / process group holder -> the sample_n sequences of the ONE shape are split over
        # its ranks (single-shape option; dist.sample_n_sharded) instead of every rank completing whole shapes
        self.shard_sample_n = shard_sample_n
        self.temperature, self.sample_n, self.top_k, self.top_p, self.depth, self.decode_res = temperature, sample_n, top_k, top_p, depth, decode_res
        self.sample_max_step, self.end_tokens = sample_max_step, tuple(end_tokens)
        self.mask_invalid, self.mask_invalid_completion = mask_invalid, mask_invalid_completion
        self.force_keep_c_indices, self.sort_prob, self.thresh = force_keep_c_indices, sort_prob, thresh
        self.keep_logits_history, self.seed = keep_logits_history, seed

    @torch.no_grad()
    def compute_batch(self, batch, input_name=""):
        m = self.pl_module                      # plugin.ShapeFormerModel
        pipe, g = m.pipe, m.transformer
        Xct = batch["Xct"].to(g.dev, torch.float32)
        assert Xct.shape[0] == 1                 # shapeformer.py:227
        enc = pipe.encode_cloud(Xct)             # representer.get_indices(stage="test"): c tokens + empty index
        Lc = int(enc["Lc"][0])
        c_ind = enc["c_tokens"][:, :Lc].long()   # origin_c_indices (1, L_c, 2), end-token pair last
        z_ind = None
        if "Xbd" in batch:
            z = pipe.encode_cloud(batch["Xbd"].to(g.dev, torch.float32))
            z_ind = z["c_tokens"][:, :int(z["Lc"][0])].long()
        S = self.sample_n
        skw = dict(max_steps=self.sample_max_step, top_k=self.top_k, top_p=self.top_p, temperature=self.temperature,
                   best_in_first=True, mask_invalid=self.mask_invalid, mask_invalid_completion=self.mask_invalid_completion, seed=self.seed)
        if self.shard_sample_n is not None and not self.keep_logits_history:
            from .dist import sample_n_sharded
            res = sample_n_sharded(g, enc["c_tokens"], enc["Lc"], S, self.shard_sample_n, **skw)     # identical tokens, S / world rows per rank
        else:
            res = g.sample(enc["c_tokens"].expand(S, -1, -1).contiguous(), enc["Lc"].expand(S).contiguous(),
                           return_logits=self.keep_logits_history,
                           shared_prefix="auto", **skw)     # the S rows are copies of ONE condition: prefill + condition K/V once where it pays
        computed = dict(batch=batch, samples=res["samples"], origin_samples=res["samples"],
                        logits_history=res.get("logits_history"), c_ind=c_ind,
                        z_ind=z_ind if z_ind is not None else c_ind[:, :0], empty_index=enc["empty_index"].long()[0])
        computed = _np(computed)
        if self.sort_prob:
            # compute_log_probs(samples, logits_history): the sampler already accumulated exactly these terms on the device
            computed["log_prob"] = res["log_prob"].numpy().astype(np.float64) if isinstance(res["log_prob"], torch.Tensor) else np.asarray(res["log_prob"], np.float64)
        return computed

    def sample_order(self, computed):
        """shapeformer.py:281-289: most probable sequence first."""
        S = computed["samples"].shape[0]
        if not self.sort_prob:
            return np.arange(S)
        lp = computed["log_prob"] if "log_prob" in computed else compute_log_probs(computed["samples"], computed["logits_history"])
        return np.argsort(np.array([x.sum() for x in lp]))[::-1]

    def _token_sets(self, computed):
        """(key, (L,2) tokens) in the reference's insertion order: data_z, data_c, then s<i> by probability."""
        c_ind, z_ind, samples = computed["c_ind"], computed["z_ind"], computed["samples"]
        sets = []
        if "Xbd" in computed["batch"]:
            sets.append(("data_z", z_ind[0]))
        sets.append(("data_c", c_ind[0]))
        for i in self.sample_order(computed):
            sample = samples[i]
            if self.force_keep_c_indices:      # shapeformer.py:293-299: condition tokens win on shared positions
                cated = np.concatenate([c_ind[0], sample], 0)
                uni, uniind = np.unique(cated[:, 0], return_index=True)
                sample = np.stack([uni, cated[uniind, 1]], 1)
            sets.append((f"s{i}", sample))
        return sets

    @torch.no_grad()
    def visualize_batch(self, computed, input_name="", data_dir=None):
        """vis_ind (shapeformer.py:332-379) for every token set, batched: tokens -> dense code grid -> occupancy on the
        decode_res^3 lattice -> mesh; then the `meshes/` and `eval/` files of shapeformer.py:303-327."""
        from . import mcubes
        data_dir = data_dir or self.data_dir
        vq = self.pl_module.representer.vqvae_model.core
        R, Q = 2 ** self.depth, self.decode_res
        sets = [(k, filter_end_tokens(t, self.end_tokens)) for k, t in self._token_sets(computed)]
        sets = [(k, t) for k, t in sets if len(t)]                           # empty sequence: blank image, no mesh
        out = {}
        if not sets:
            return out
        dense = np.full((len(sets), R ** 3), int(computed["empty_index"]), np.int32)
        for j, (_, t) in enumerate(sets):
            dense[j, t[:, 0]] = t[:, 1]                                      # batch_sparse2dense (common.py:171-189)
        occ = vq.decode_index(torch.from_numpy(dense.reshape(-1, R, R, R)).to(vq.dev), grid_Q=Q, sigmoid=True)["logits"]
        v, f, voff, toff = mcubes.marching_cubes_dev(occ.reshape(len(sets), Q, Q, Q), self.thresh)
        v, f = v.cpu().numpy().astype(np.float64), f.cpu().numpy().astype(int)
        eval_pcs = []
        for j, (key, _) in enumerate(sets):
            vert, face = v[voff[j]:voff[j + 1]], f[toff[j]:toff[j + 1]]
            if vert.shape[0] < 10:                                           # shapeformer.py:316-317
                continue
            path = os.path.join(data_dir, "meshes", f"{input_name}_{key}_mesh.ply")
            meshio.write_ply(path, vert, face)
            out[key + "_mesh"] = {"vert": vert, "face": face, "path": path}
            if key[0] == "s":
                eval_pcs.append(meshio.sample_mesh(vert, face, 10 ** 5))
        if eval_pcs:
            os.makedirs(os.path.join(data_dir, "eval"), exist_ok=True)
            ed = dict(eval_pc=eval_pcs[0])
            for i, pc in enumerate(eval_pcs):
                ed[f"recon_{i}"] = pc
            np.savez(os.path.join(data_dir, "eval", f"{input_name}.npz"), **ed)
        return out
