"""GPU: the data side feeding the device path: DataModule.batches / collate_to_device put dict batches of the
reference's item schema on the HIP device (pinned host staging, one async H2D copy per key) and the completion / training
entry points consume them unchanged."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "data_side.npz"))


def test_datamodule_batches_on_device_feed_the_hot_path(dev, tmp_path):
    from shapeformer_amd import data as D
    from shapeformer_amd.vqdif import VQDIF
    rs = np.random.RandomState(0)
    d = tmp_path / "datasets" / "IMNet2_64" / "train"
    d.mkdir(parents=True)
    clouds = np.stack([G["cloud"][rs.choice(3000, 2000)] * s for s in (1.0, 0.8, 0.6, 0.9)])
    np.save(d / "Xbd.npy", clouds)
    np.save(d / "Ytg.npy", np.packbits(rs.rand(4, 512) > 0.5, axis=-1))
    np.save(d / "cate_5.npy", np.array([2, 0, 3]))
    kw = dict(dataset="IMNet2_64", split="train", boundary_N=1024, target_N=64, grid_dim=8, root=str(tmp_path / "datasets"), cate="all",
              partial_opt={"class": "shapeformer.data.partial.VirtualScanSelector", "kwargs": {"context_N": 512}})
    opt = {"class": "shapeformer.data.imnet_datasets.imnet_datasets.Imnet2LowResDataset", "kwargs": kw}
    dm = D.DataModule(batch_size=2, num_workers=0, trainset_opt=opt, testset_opt=opt)
    dm.setup()
    np.random.seed(7)
    want = [dm.train_set[i] for i in range(4)]                    # host items under the seed
    np.random.seed(7)
    got = list(dm.batches("train", dev))                          # the same draws, batched on the device
    assert len(got) == 2
    for bi, b in enumerate(got):
        assert set(b) == {"Xct", "Xbd", "Xtg", "Ytg"}
        for k, v in b.items():
            assert v.device.type == "cuda" and v.dtype == torch.float32 and v.shape[0] == 2
            assert np.array_equal(v.cpu().numpy(), np.stack([want[2 * bi + j][k] for j in range(2)]))
    # explicit collate + index subset (the inference drivers' visual_indices route), consumed by the encoder as is
    b = D.collate_to_device([dm.test_set[i] for i in (3, 1)], dev, keys=["Xct", "Xbd"])
    assert set(b) == {"Xct", "Xbd"} and b["Xbd"].shape == (2, 1024, 3)
    vq = VQDIF(res=16, device=dev)
    q, mode, enc = vq.quantize_cloud(b["Xct"])
    assert q.shape == (2, 16, 16, 16) and bool(enc["grid_mask"].any())
