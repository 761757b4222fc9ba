"""Data side of the path: what produces the {Xct, Xbd, Xtg, Ytg} items the hot path consumes.

Mirrors, with the reference's ctor kwargs and the reference's ORDER of numpy-RNG draws (so a seeded run selects the
same points):

  sample_sphere / hidden_point_removal   xgutils/geoutil.py:45-74   (Katz et al. HPR: spherical flip + convex hull)
  AllSelector / BallSelector / MultiBallSelector / VirtualScanSelector   shapeformer/data/partial.py:60-146
  apply_random_transforms / TransformDataset   shapeformer/data/paper_datasets/transform_dataset.py:19-112
  ListDataset                             shapeformer/data/paper_datasets/list_dataset.py:14-41
  Imnet2LowResDataset                     shapeformer/data/imnet_datasets/imnet_datasets.py:144-224 (HDF5: `Xbd` clouds,
                                          bit-packed `Ytg` occupancy, `cate_<k>` index lists)
  DataModule                              shapeformer/datamodule.py:13-63 (plain loaders, no Lightning)
  collate_to_device                       GPU-side batching: stack on the host once, one H2D copy per key

This is host-side numpy/scipy like the reference (its DataLoader workers never touch the GPU); the device work starts
at `collate_to_device`.  h5py is not part of this image: Imnet2LowResDataset imports it lazily and also accepts the
same arrays as a directory of .npy files (`<dataset>/<split>/{Xbd,Ytg,cate_*}.npy`).
"""
from __future__ import annotations

import copy
import os

import numpy as np
import torch


# ----------------------------------------------------------------------------- geometry helpers
def sample_sphere(point_N, dim=3):
    v = np.random.randn(point_N, dim)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def spherical_flip(points, center, param):
    """Katz, Tal, Basri 2007: reflect every point about the sphere of radius max|p| * 10^param around `center`."""
    p = points - center
    n = np.linalg.norm(p, axis=1)
    R = n.max() * np.power(10.0, param)
    return p + 2.0 * (R - n)[:, None] * p / n[:, None]


def hidden_point_removal(cloud, campos):
    """Points of `cloud` visible from `campos`: vertices of the convex hull of the flipped cloud plus the viewpoint.
    Like the reference, the LAST hull vertex (the appended viewpoint, hull vertices come in input order) is dropped."""
    from scipy.spatial import ConvexHull
    flipped = spherical_flip(cloud, np.asarray(campos, float)[None, :], np.pi)
    hull = ConvexHull(np.concatenate([flipped, np.zeros((1, cloud.shape[1]))], 0))
    return cloud[hull.vertices[:-1]]


# ----------------------------------------------------------------------------- partial-cloud selectors
def _resample(X, n):
    return X[np.random.choice(X.shape[0], n, replace=True)]


def _jitter(X, noise):
    return (X + np.random.randn(*X.shape) * noise).clip(-1., 1.)


class AllSelector:
    def __init__(self, context_N=None):
        self.context_N = context_N

    def __call__(self, Xbd, **_):
        return Xbd if self.context_N is None else _resample(Xbd, self.context_N)


class BallSelector:
    """Points within `radius` of a random pivot point (or outside it when inverse=True; fewer than 400 left -> all)."""

    def __init__(self, radius=.1, context_N=512, noise=0., inverse=False):
        self.radius, self.context_N, self.noise, self.inverse = radius, context_N, noise, inverse

    def __call__(self, Xbd, radius=None, **_):
        pivot = Xbd[np.random.choice(Xbd.shape[0], 1)]
        d2 = np.square(Xbd - pivot).sum(-1)
        keep = d2 < (self.radius if radius is None else radius) ** 2
        if self.inverse:
            keep = ~keep
            if keep.sum() < 400:
                keep = np.ones_like(keep)
        Xct = Xbd[keep]
        if self.context_N >= 0:
            Xct = _resample(Xct, self.context_N)
        return _jitter(Xct, self.noise) if self.noise > 0 else Xct


class VirtualScanSelector:
    """What a depth camera on a sphere of `radius` around the object sees (hidden point removal)."""

    def __init__(self, radius=10, context_N=512, noise=0., manual_cameras=None):
        self.radius, self.context_N, self.noise, self.manual_cameras = radius, context_N, noise, manual_cameras or {}

    def __call__(self, Xbd, index=None, **_):
        cam = sample_sphere(1)[0] * self.radius
        Xct = hidden_point_removal(Xbd, cam)
        if Xct.shape[0] <= 2:
            Xct = Xbd
        if self.context_N >= 0:
            Xct = _resample(Xct, self.context_N)
        return _jitter(Xct, self.noise) if self.noise > 0 else Xct


class MultiBallSelector:
    def __init__(self, radius_range=(.05, .4), N_range=(1, 3), context_N=512, virtual_scan=False):
        self.radius_range, self.N_range, self.context_N, self.virtual_scan = radius_range, N_range, context_N, virtual_scan
        self.selector = BallSelector(context_N=context_N)

    def __call__(self, Xbd, **_):
        n = np.random.randint(*self.N_range)
        lo, hi = self.radius_range
        if self.virtual_scan:
            seen = hidden_point_removal(Xbd, sample_sphere(1)[0] * 10)
            Xbd = seen if seen.shape[0] > 2 else Xbd
        parts = []
        for _ in range(n):
            r = lo + np.random.rand() * (hi - lo)
            parts.append(self.selector(Xbd, radius=r))
        return _resample(np.concatenate(parts, 0), self.context_N)


# ----------------------------------------------------------------------------- augmentation
def occupied_voxels(points, grid_dim):
    """ptutil.point2voxel(...).sum(): number of distinct cells of the [-1,1]^3 `grid_dim` grid hit by `points`
    (index = clamp(round((p+1)/2*grid_dim - .5), 0, grid_dim-1), ptutil.py:445-450; float32 like the reference)."""
    p = torch.from_numpy(np.ascontiguousarray(points))          # float64 in, float64 arithmetic (as the reference calls it)
    idx = torch.clamp(torch.round((p + 1) / 2 * grid_dim - 0.5), 0.0, grid_dim - 1).long()
    flat = (idx[:, 0] * grid_dim + idx[:, 1]) * grid_dim + idx[:, 2]
    return int(torch.unique(flat).numel())


def apply_random_transforms(X, Ys, mode=(), max_voxels=812, voxel_dim=16):
    """transform_dataset.py:46-84: normalise X's bounding box to [-.3,.3] (longest side .6), then random
    rot_axis_y / rot / scale / shift of every array in Ys; shrink if more than `max_voxels` cells would be occupied."""
    from scipy.spatial.transform import Rotation
    hi, lo = X.max(0), X.min(0)
    center, span = (hi + lo) / 2, (hi - lo).max()
    Xn = (X - center) / span * .6
    Ys = {k: (v - center) / span * .6 for k, v in Ys.items()}
    if "rot_axis_y" in mode:
        r = Rotation.from_rotvec(np.random.rand() * 2 * np.pi * np.array([0., 1., 0.]))
        Xn, Ys = r.apply(Xn), {k: r.apply(v) for k, v in Ys.items()}
    if "rot" in mode:
        r = Rotation.random()
        Xn, Ys = r.apply(Xn), {k: r.apply(v) for k, v in Ys.items()}
    if "scale" in mode:
        s = 1 + np.random.rand(1)[0] * (0.99 / np.abs(Xn).max() - 1)
        Xn, Ys = Xn * s, {k: v * s for k, v in Ys.items()}
    n_vox = np.float32(occupied_voxels(Xn, voxel_dim))
    if n_vox > max_voxels:
        s = (max_voxels / float(n_vox)) ** (2 / 3.)
        Xn, Ys = Xn * s, {k: v * s for k, v in Ys.items()}
    if "shift" in mode:
        hi, lo = Xn.max(0), Xn.min(0)
        shift = np.random.rand(1, Xn.shape[-1]) * ((1 - hi) - (-1 - lo)) + (-1 - lo)
        Xn, Ys = Xn + shift, {k: v + shift for k, v in Ys.items()}
    return Ys


class TransformDataset:
    def __init__(self, split="test", mode=("rot_axis_y", "scale"), apply_Xtg=False, max_voxels=100, voxel_dim=16, dset_opt=None):
        self.split, self.mode, self.apply_Xtg, self.max_voxels, self.voxel_dim = split, list(mode), apply_Xtg, max_voxels, voxel_dim
        self.dset = instantiate(dset_opt or {})

    def __len__(self):
        return len(self.dset)

    def __getitem__(self, ind):
        item = self.dset[ind]
        if "Xbd" in item:
            ys = {k: item[k].copy() for k in ("Xbd", "Xct") if k in item}
            if self.apply_Xtg and "Xtg" in item:
                ys["Xtg"] = item["Xtg"].copy()
            for k, v in apply_random_transforms(item["Xbd"].copy(), ys, self.mode, self.max_voxels, self.voxel_dim).items():
                item[k] = v.astype(np.float32)
        return item


# ----------------------------------------------------------------------------- datasets
class ListDataset:
    """Items are directories `<dir of list file>/<name>/{Xbd,Xct}.npy`; subsampled to boundary_N / context_N points."""

    def __init__(self, ditem_list, split="test", load_keys=("Xbd", "Xct"), subsample=True, boundary_N=32768, context_N=16384,
                 evalseed=314, **_):
        self.names = np.atleast_1d(np.loadtxt(ditem_list, dtype=str))
        self.root, self.load_keys, self.subsample = os.path.dirname(ditem_list), list(load_keys), subsample
        self.boundary_N, self.context_N = boundary_N, context_N

    def __len__(self):
        return len(self.names)

    def __getitem__(self, ind):
        d = os.path.join(self.root, str(self.names[ind]))
        item = {k: np.load(os.path.join(d, f"{k}.npy")) for k in self.load_keys}
        if self.subsample:
            for k, n in (("Xbd", self.boundary_N), ("Xct", self.context_N)):
                if k in item:
                    item[k] = item[k][np.random.choice(item[k].shape[0], n)]
        return item


def make_grid(lo, hi, shape):
    """nputil.makeGrid(..., indexing='ij') 'on' mode: linspace per axis, first axis slowest -> (prod(shape), dim)."""
    axes = [np.linspace(l, h, n) for l, h, n in zip(lo, hi, shape)]
    return np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, len(shape))


class _ArrayStore:
    """`f[name][index]` over an HDF5 file (h5py, opened per access like nputil.H5Var) or a directory of .npy files."""

    def __init__(self, path):
        self.path = path
        self.h5 = os.path.isfile(path)
        if not self.h5 and not os.path.isdir(os.path.splitext(path)[0]):
            raise FileNotFoundError(f"{path}: neither an HDF5 file nor a directory {os.path.splitext(path)[0]}/ of .npy arrays")
        self.dir = os.path.splitext(path)[0]
        self._mm = {}

    def get(self, name, index=None):
        if self.h5:
            import h5py     # not in this image; present wherever the reference's datasets are
            with h5py.File(self.path, "r") as f:
                return np.array(f[name]) if index is None else f[name][index]
        if name not in self._mm:
            self._mm[name] = np.load(os.path.join(self.dir, name + ".npy"), mmap_mode="r")
        a = self._mm[name]
        return np.array(a) if index is None else np.array(a[index])

    def length(self, name):
        if self.h5:
            import h5py
            with h5py.File(self.path, "r") as f:
                return f[name].shape[0]
        return self.get(name, None).shape[0] if name not in self._mm else self._mm[name].shape[0]


def balanced_sampling2(Xbd, Xtg, Ytg, target_N=4096, x_dim=3, random_scale=.1):
    """imnet_datasets.py:288-304 (`weighted_sampling=True`), numpy draw for draw: target_N//2 boundary-point INDICES, a jitter
    draw of the same count (the reference jitters those boundary points and looks their cells up, but returns neither), then
    target_N//2 lattice indices; the sample is Xtg / Ytg at the concatenated index list - i.e. the boundary indices address the
    TARGET lattice (as the reference's `Xtg[choice]` does; its torch.randint draw uses torch's generator and is unused)."""
    rdc_xbd = np.random.choice(Xbd.shape[0], target_N // 2, replace=True)
    np.random.randn(len(rdc_xbd), x_dim)
    rdc1 = np.random.choice(Xtg.shape[0], target_N // 2, replace=True)
    choice = np.concatenate([rdc_xbd, rdc1])
    return Xtg[choice], Ytg[choice]


class Imnet2LowResDataset:
    def __init__(self, dataset="IMNet2_64", cate="all", zoomfac=1, duplicate_size=1, split="train", boundary_N=2048,
                 target_N=-1, grid_dim=64, weighted_sampling=False, Xbd_as_Xct=False, Xct_as_Xbd=False, partial_opt=None,
                 root="datasets"):
        self.weighted_sampling = bool(weighted_sampling)
        self.store = _ArrayStore(os.path.join(root, dataset, f"{split}.hdf5"))
        n = self.store.length("Xbd")
        if isinstance(cate, str):
            self.subset = np.arange(n) if cate == "all" else self.store.get(f"cate_{cate}")
        else:
            self.subset = np.concatenate([self.store.get(f"cate_{c}") for c in cate])
        self.length = len(self.subset)
        self.duplicate_size = duplicate_size if split == "train" else 1
        self.boundary_N, self.target_N, self.grid_dim = boundary_N, target_N, grid_dim
        self.Xbd_as_Xct, self.Xct_as_Xbd = Xbd_as_Xct, Xct_as_Xbd
        self.partial_selector = instantiate(partial_opt or {"class": "shapeformer.data.partial.BallSelector",
                                                            "kwargs": dict(radius=.4, context_N=512)})
        self.all_Xtg = make_grid([-1, -1, -1.], [1., 1, 1], [grid_dim] * 3)

    def __len__(self):
        return self.length * self.duplicate_size

    def __getitem__(self, index, all_target=False):
        o_ind = index % self.length
        src = self.subset[o_ind]
        Xbd = self.store.get("Xbd", src)
        Xct = np.float32(Xbd if self.Xbd_as_Xct else self.partial_selector(Xbd, index=o_ind))
        Xbd = Xbd[np.random.choice(Xbd.shape[0], self.boundary_N, replace=True)]
        Ytg = np.unpackbits(self.store.get("Ytg", src), axis=-1)[..., None]      # bit-packed occupancy of the grid_dim^3 lattice
        Xtg = self.all_Xtg
        if self.weighted_sampling:       # imnet_datasets.py:196-203
            Xtg, Ytg = balanced_sampling2(Xbd, Xtg, Ytg, target_N=self.target_N if self.target_N != -1 else Xtg.shape[0], x_dim=Xbd.shape[-1])
        elif self.target_N != -1 and not all_target:
            pick = np.random.choice(Xtg.shape[0], self.target_N, replace=True)
            Xtg, Ytg = Xtg[pick], Ytg[pick]
        if self.Xct_as_Xbd:
            Xbd = Xct
        return dict(Xct=Xct.astype(np.float32), Xbd=Xbd.astype(np.float32), Xtg=Xtg.astype(np.float32), Ytg=Ytg.astype(np.float32))


# ----------------------------------------------------------------------------- plugin glue + loaders
DATA_REGISTRY = {
    "shapeformer.data.partial.AllSelector": AllSelector,
    "shapeformer.data.partial.BallSelector": BallSelector,
    "shapeformer.data.partial.MultiBallSelector": MultiBallSelector,
    "shapeformer.data.partial.VirtualScanSelector": VirtualScanSelector,
    "shapeformer.data.paper_datasets.transform_dataset.TransformDataset": TransformDataset,
    "shapeformer.data.paper_datasets.list_dataset.ListDataset": ListDataset,
    "shapeformer.data.imnet_datasets.imnet_datasets.Imnet2LowResDataset": Imnet2LowResDataset,
}


def instantiate(opt):
    """sysutil.instantiate_from_opt for the data-side classes (None when `class` is missing / None)."""
    if not opt or opt.get("class") is None:
        return None
    cls = DATA_REGISTRY.get(opt["class"])
    if cls is None:
        from .plugin import load_object
        cls = load_object(opt["class"])
    return cls(**opt.get("kwargs", {}))


def collate_to_device(items, device, keys=None):
    """GPU-side batching: stack each key once on the host (pinned when a HIP device is present) and issue one
    asynchronous H2D copy per key -> dict of (B, ...) float32 device tensors."""
    keys = keys or [k for k, v in items[0].items() if isinstance(v, np.ndarray)]
    out = {}
    for k in keys:
        t = torch.from_numpy(np.stack([np.asarray(it[k], np.float32) for it in items]))
        if torch.device(device).type == "cuda" and torch.cuda.is_available():
            t = t.pin_memory()
        out[k] = t.to(device, non_blocking=True)
    return out


class DataModule:
    """datamodule.py:13-63 without Lightning: `setup()` instantiates the sets, `*_dataloader()` return torch DataLoaders
    of dict batches (numpy collation to tensors), `batches(split, device)` yields device-resident batches."""

    def __init__(self, batch_size=32, test_batch_size=None, val_batch_size=None, num_workers=8, trainset_opt=None,
                 valset_opt=None, testset_opt=None, visualset_opt=None):
        empty = {"class": None, "kwargs": {}}
        self.opts = {k: copy.deepcopy(v or empty) for k, v in dict(train=trainset_opt, val=valset_opt, test=testset_opt,
                                                                  visual=visualset_opt).items()}
        for k in ("train", "val", "test"):
            self.opts[k].setdefault("kwargs", {}).setdefault("split", k)
        self.batch_size, self.num_workers = batch_size, num_workers
        self.test_batch_size = test_batch_size if test_batch_size is not None else batch_size
        self.val_batch_size = val_batch_size if val_batch_size is not None else test_batch_size

    def setup(self, stage=None):
        self.train_set = self.val_set = self.test_set = None
        if stage in ("fit", "train", "val", None):
            self.train_set, self.val_set = instantiate(self.opts["train"]), instantiate(self.opts["val"])
        if stage == "test" or stage is None or self.val_set is None or self.opts["test"]["class"] is not None:
            self.test_set = instantiate(self.opts["test"])
        if self.opts["val"]["class"] is None:
            self.val_set, self.val_batch_size = self.test_set, self.test_batch_size
        self.visual_set = self.val_set if self.opts["visual"]["class"] is None else instantiate(self.opts["visual"])

    def _loader(self, ds, bs, shuffle, workers=None):
        from torch.utils.data import DataLoader
        return DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=self.num_workers if workers is None else workers)

    def train_dataloader(self, shuffle=True):
        return self._loader(self.train_set, self.batch_size, shuffle)

    def val_dataloader(self, shuffle=False):
        return self._loader(self.val_set, self.val_batch_size, False)

    def test_dataloader(self, shuffle=False):
        return self._loader(self.test_set, self.test_batch_size, False)

    def visual_dataloader(self, shuffle=False):
        return self._loader(self.visual_set, 1, False, workers=1)

    def batches(self, split, device, batch_size=None, indices=None):
        ds = getattr(self, f"{split}_set")
        bs = batch_size or (self.batch_size if split == "train" else self.test_batch_size)
        idx = list(range(len(ds))) if indices is None else list(indices)
        for i in range(0, len(idx), bs):
            yield collate_to_device([ds[j] for j in idx[i:i + bs]], device)
