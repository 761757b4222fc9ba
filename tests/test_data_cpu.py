"""Data side: shapeformer_amd.data replayed under the seeds of oracle/make_golden_data.py must select
the points the REAL reference code selected (fixture tests/golden/data_side.npz), plus dataset / loader plumbing."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = np.load(os.path.join(ROOT, "tests", "golden", "data_side.npz"))


def test_hidden_point_removal_and_selectors_match_reference_draw_for_draw():
    from shapeformer_amd import data as D
    X = G["cloud"]
    np.random.seed(11)
    cam = D.sample_sphere(1)[0] * 10
    assert np.array_equal(cam, G["hpr_cam"])
    assert np.array_equal(D.hidden_point_removal(X, cam), G["hpr"])
    cases = {
        "ball": (D.BallSelector, dict(radius=.4, context_N=512)),
        "ball_inv_noise": (D.BallSelector, dict(radius=.3, context_N=700, noise=0.01, inverse=True)),
        "vscan": (D.VirtualScanSelector, dict(context_N=2048)),
        "multiball": (D.MultiBallSelector, dict(context_N=600, virtual_scan=True)),
        "all": (D.AllSelector, dict(context_N=300)),
    }
    for i, (name, (cls, kw)) in enumerate(cases.items()):
        np.random.seed(100 + i)
        got = cls(**kw)(X.copy())
        assert got.shape == G["sel_" + name].shape and np.array_equal(got, G["sel_" + name]), name


def test_random_transforms_match_reference():
    from shapeformer_amd import data as D
    X = G["cloud"]
    for i, mode in enumerate((["scale"], ["rot_axis_y", "scale"], ["rot", "scale", "shift"], ["scale"])):
        np.random.seed(200 + i)
        Ys = D.apply_random_transforms(X.copy(), {"Xbd": X.copy(), "Xct": X[:500].copy()}, mode=mode,
                                       max_voxels=40 if i == 3 else 512, voxel_dim=16)
        for k in ("Xbd", "Xct"):
            assert np.abs(Ys[k] - G[f"tf{i}_{k}"]).max() < 1e-6, (i, k)      # fixture stored as float32
        assert np.abs(Ys["Xbd"]).max() <= (1.0 if "shift" in mode else 0.99) + 1e-9
    assert D.occupied_voxels(G["tf3_Xbd"].astype(np.float64), 16) <= 48       # the shrink branch brought the count down


def test_datasets_datamodule_and_device_batching(tmp_path):
    from shapeformer_amd import data as D
    rs = np.random.RandomState(0)
    # IMNet2_64-style store as .npy arrays (h5py is not in this image): clouds, bit-packed 8^3 occupancy, category lists
    d = tmp_path / "datasets" / "IMNet2_64" / "train"
    d.mkdir(parents=True)
    clouds = np.stack([G["cloud"][rs.choice(3000, 2000)] * s for s in (1.0, 0.8, 0.6)])
    occ = rs.rand(3, 512) > 0.5
    np.save(d / "Xbd.npy", clouds)
    np.save(d / "Ytg.npy", np.packbits(occ, axis=-1))
    np.save(d / "cate_5.npy", np.array([2, 0]))
    kw = dict(dataset="IMNet2_64", split="train", boundary_N=256, target_N=64, grid_dim=8, root=str(tmp_path / "datasets"),
              partial_opt={"class": "shapeformer.data.partial.VirtualScanSelector", "kwargs": {"context_N": 128}})
    ds = D.Imnet2LowResDataset(cate="5", **kw)
    assert len(ds) == 2
    np.random.seed(3)
    it = ds[0]
    assert {k: v.shape for k, v in it.items()} == {"Xct": (128, 3), "Xbd": (256, 3), "Xtg": (64, 3), "Ytg": (64, 1)}
    assert all(v.dtype == np.float32 for v in it.values())
    full = ds.__getitem__(1, all_target=True)                 # subset[1] == item 0: all 8^3 targets, unpacked bits
    assert np.array_equal(full["Ytg"][:, 0], occ[0].astype(np.float32))
    assert np.allclose(full["Xtg"], D.make_grid([-1, -1, -1], [1, 1, 1], [8, 8, 8]))
    assert np.allclose(full["Xtg"][1], [-1, -1, -1 + 2 / 7])   # 'ij': last axis fastest
    # weighted_sampling=True (imnet_datasets.py:196-203): half of the targets at boundary-point INDICES of the lattice, half uniform
    dw = D.Imnet2LowResDataset(cate="5", weighted_sampling=True, **kw)
    np.random.seed(3)
    iw = dw[0]
    assert iw["Xtg"].shape == (64, 3) and iw["Ytg"].shape == (64, 1) and np.array_equal(iw["Xct"], it["Xct"]) and np.array_equal(iw["Xbd"], it["Xbd"])
    # TransformDataset on top, resolved from the reference dotted names (datamodule YAML layout)
    opt = {"class": "shapeformer.data.paper_datasets.transform_dataset.TransformDataset",
           "kwargs": dict(max_voxels=512, voxel_dim=16, mode=["scale"],
                          dset_opt={"class": "shapeformer.data.imnet_datasets.imnet_datasets.Imnet2LowResDataset",
                                    "kwargs": dict(cate="all", **kw)})}
    dm = D.DataModule(batch_size=2, num_workers=0, trainset_opt=opt, testset_opt=opt)
    dm.setup()
    assert len(dm.train_set) == 3 and dm.val_set is dm.test_set
    b = next(iter(dm.train_dataloader(shuffle=False)))
    assert b["Xbd"].shape == (2, 256, 3) and b["Xct"].shape == (2, 128, 3) and float(b["Xbd"].abs().max()) <= 0.99 + 1e-6
    bb = next(dm.batches("train", "cpu", batch_size=3))
    assert bb["Xtg"].shape == (3, 64, 3) and bb["Xtg"].dtype.is_floating_point
    # ListDataset: <list dir>/<name>/{Xbd,Xct}.npy
    for n in ("a", "b"):
        (tmp_path / "demo" / n).mkdir(parents=True)
        np.save(tmp_path / "demo" / n / "Xbd.npy", clouds[0])
        np.save(tmp_path / "demo" / n / "Xct.npy", clouds[1][:500])
    (tmp_path / "demo" / "list.txt").write_text("a\nb\n")
    ld = D.instantiate({"class": "shapeformer.data.paper_datasets.list_dataset.ListDataset",
                        "kwargs": dict(ditem_list=str(tmp_path / "demo" / "list.txt"), boundary_N=100, context_N=50)})
    assert len(ld) == 2 and ld[1]["Xbd"].shape == (100, 3) and ld[1]["Xct"].shape == (50, 3)


def test_weighted_target_sampling_replays_the_reference_draw_for_draw():
    """imnet_datasets.py:288-304 (`weighted_sampling=True` -> balanced_sampling2): the same numpy seed must select the same
    target points / labels as the REAL reference did (tests/golden/data_side_ws.npz, oracle/make_golden_data.py) and leave the
    numpy stream at the same position; the dataset takes the route when the flag is set."""
    from shapeformer_amd import data as D
    W = np.load(os.path.join(ROOT, "tests", "golden", "data_side_ws.npz"))
    Xtg = D.make_grid([-1, -1, -1.], [1., 1, 1], [16] * 3)
    np.random.seed(300)
    sx, sy = D.balanced_sampling2(W["Xbd"], Xtg, W["Ytg"], target_N=96, x_dim=3)
    assert np.array_equal(sx, W["sub_Xtg"]) and np.array_equal(sy, W["sub_Ytg"])
    assert np.random.rand() == float(W["next_rand"])
