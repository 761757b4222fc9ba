"""ORACLE tooling (this container only): pins shapeformer_amd/data.py against the REAL reference data-side code.

Imports shapeformer/data/partial.py, xgutils/geoutil.py's hidden_point_removal and
shapeformer/data/paper_datasets/{transform_dataset,list_dataset}.py from /root/reference (with the stub modules of
oracle/refimport.py), runs them under fixed numpy seeds on a synthetic cloud and writes inputs + outputs to
tests/golden/data_side.npz.  tests/test_data_cpu.py replays the same seeds through shapeformer_amd.data.

    python oracle/make_golden_data.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402


def cloud(seed, n=3000):
    rs = np.random.RandomState(seed)
    u = rs.randn(n, 3)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    p = u * np.array([0.55, 0.3, 0.42]) + 0.08 * np.sin(7 * u[:, [1, 2, 0]])   # a bumpy ellipsoid surface
    return p.astype(np.float64)


def main():
    assert refimport.reference_available()
    refimport.setup()
    partial = importlib.import_module("shapeformer.data.partial")
    td = importlib.import_module("shapeformer.data.paper_datasets.transform_dataset")
    geoutil = importlib.import_module("xgutils.geoutil")
    X = cloud(1)
    out = {"cloud": X}
    np.random.seed(11)
    out["hpr_cam"] = geoutil.sample_sphere(1)[0] * 10
    out["hpr"] = geoutil.hidden_point_removal(X, out["hpr_cam"])
    cases = {
        "ball": (partial.BallSelector, dict(radius=.4, context_N=512)),
        "ball_inv_noise": (partial.BallSelector, dict(radius=.3, context_N=700, noise=0.01, inverse=True)),
        "vscan": (partial.VirtualScanSelector, dict(context_N=2048)),
        "multiball": (partial.MultiBallSelector, dict(context_N=600, virtual_scan=True)),
        "all": (partial.AllSelector, dict(context_N=300)),
    }
    for i, (name, (cls, kw)) in enumerate(cases.items()):
        np.random.seed(100 + i)
        out["sel_" + name] = np.asarray(cls(**kw)(X.copy()))
    for i, mode in enumerate((["scale"], ["rot_axis_y", "scale"], ["rot", "scale", "shift"], ["scale"])):
        np.random.seed(200 + i)
        mv = 40 if i == 3 else 512                      # i == 3 forces the max_voxels shrink branch
        Ys = td.apply_random_transforms(X.copy(), {"Xbd": X.copy(), "Xct": X[:500].copy()}, mode=mode, max_voxels=mv, voxel_dim=16)
        out[f"tf{i}_Xbd"], out[f"tf{i}_Xct"] = Ys["Xbd"].astype(np.float32), Ys["Xct"].astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "data_side.npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote", path, {k: np.asarray(v).shape for k, v in out.items()})
    # weighted target sampling (imnet_datasets.py:288-304, `weighted_sampling=True`): the reference's own function under a fixed numpy
    # seed; a separate fixture so that data_side.npz stays byte-identical
    imd = importlib.import_module("shapeformer.data.imnet_datasets.imnet_datasets")
    from shapeformer_amd.data import make_grid
    Xtg = make_grid([-1, -1, -1.], [1., 1, 1], [16] * 3)
    Ytg = (np.linalg.norm(Xtg - 0.1, axis=-1, keepdims=True) < 0.6).astype(np.uint8)
    Xb = X[:2048]
    np.random.seed(300)
    sx, sy = imd.balanced_sampling2(Xb, Xtg, Ytg, target_N=96, x_dim=3, grid_dim=16)
    after = np.random.rand()          # the numpy stream position after the call must match as well
    path2 = os.path.join(ROOT, "tests", "golden", "data_side_ws.npz")
    np.savez_compressed(path2, Xbd=Xb, Ytg=Ytg, sub_Xtg=np.asarray(sx), sub_Ytg=np.asarray(sy), next_rand=np.float64(after))
    print("wrote", path2, np.asarray(sx).shape, np.asarray(sy).shape)


if __name__ == "__main__":
    main()
