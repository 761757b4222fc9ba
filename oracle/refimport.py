"""Import the *real* reference (QhelDIV/ShapeFormer) in THIS container only.

TEST INFRASTRUCTURE — never imported by the product path (shapeformer_amd/).
Used by oracle/make_golden.py to (a) pin the CPU oracle (oracle/*.py) against
the reference's own code at full size and (b) generate the committed golden
vectors under tests/golden/.  `/root/reference` does not exist on the GPU box;
everything here is guarded by `reference_available()`.

Recipe = SURVEY.md Appendix B: stub the missing third-party modules
(pytorch_lightning, torch_scatter, h5py, igl, mcubes, skimage, fresnel, wandb,
open3d, bashlex, pathos), put /root/reference on sys.path, build the models
straight from the shipped YAMLs.  No reference source is copied.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = os.environ.get("SHAPEFORMER_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "shapeformer", "models"))


class _AutoModule(types.ModuleType):
    """Module whose every missing attribute is another auto-module / callable."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        child = _AutoModule(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        return _AutoModule(self.__name__ + "()")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


def _install_stubs():
    import torch
    import torch.nn as nn

    for name in ("h5py", "igl", "mcubes", "skimage", "skimage.measure", "skimage.color",
                 "skimage.morphology", "fresnel", "wandb", "open3d", "bashlex", "pathos",
                 "pathos.multiprocessing", "fresnel.interact", "PIL", "PIL.Image", "matplotlib",
                 "matplotlib.pyplot", "matplotlib.cm", "matplotlib.colors", "mpl_toolkits",
                 "mpl_toolkits.mplot3d", "trimesh", "cv2", "imageio", "seaborn", "plotly",
                 "sklearn.manifold", "point_cloud_utils"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _AutoModule(name)

    # pytorch_lightning: just enough for the model classes to construct
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        global_rank = 0

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    class Callback:
        def __init__(self, *a, **k):
            pass

    class LightningDataModule:
        def __init__(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.Callback = Callback
    pl.LightningDataModule = LightningDataModule
    pl.Trainer = object
    pl.seed_everything = lambda *a, **k: None
    cb = _AutoModule("pytorch_lightning.callbacks")
    cb.Callback = Callback
    pl.callbacks = cb
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.callbacks"] = cb
    for sub in ("loggers", "utilities", "plugins", "utilities.distributed"):
        sys.modules["pytorch_lightning." + sub] = _AutoModule("pytorch_lightning." + sub)

    # torch_scatter 2.0.7 semantics restated
    ts = types.ModuleType("torch_scatter")

    def scatter_max(src, index, dim=-1, out=None, dim_size=None):
        index = index.expand_as(src)
        size = list(src.shape)
        size[dim] = int(dim_size) if dim_size is not None else int(index.max()) + 1
        res = torch.full(size, float("-inf"), dtype=src.dtype, device=src.device)
        res = res.scatter_reduce(dim, index, src, reduce="amax", include_self=True)
        res = torch.where(torch.isinf(res) & (res < 0), torch.zeros_like(res), res)
        return res, None

    def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
        index = index.expand_as(src)
        if out is None:
            size = list(src.shape)
            size[dim] = int(dim_size) if dim_size is not None else int(index.max()) + 1
            out = torch.zeros(size, dtype=src.dtype, device=src.device)
        out.scatter_add_(dim, index, src)
        cnt = torch.zeros_like(out).scatter_add_(dim, index, torch.ones_like(src))
        out.div_(cnt.clamp(min=1))
        return out

    ts.scatter_max = scatter_max
    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts


_READY = False


def setup():
    global _READY
    if _READY:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _READY = True


def load_yaml(rel):
    import yaml
    with open(os.path.join(REF_ROOT, rel)) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def load_hash_weights(module, prefix_strip: str = ""):
    """load_state_dict(strict=True) with shapeformer_amd.weights tensors keyed by name."""
    import torch
    from shapeformer_amd import weights as W
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        name = k[len(prefix_strip):] if prefix_strip and k.startswith(prefix_strip) else k
        new[k] = torch.from_numpy(W.make_tensor(name, tuple(v.shape))).to(v.dtype)
    module.load_state_dict(new, strict=True)
    return module


def build_vqdif(res: int = 16):
    """Reference VQDIF from the shipped YAML, hash weights, eval mode."""
    setup()
    from shapeformer.models.vqdif.vqdif import VQDIF
    y = load_yaml(f"configs/vqdif/shapenet_res{res}.yaml")
    m = VQDIF(**y["pl_model_opt"]["kwargs"])
    load_hash_weights(m)
    return m.eval()


def build_gpt(**overrides):
    """Reference CondTupleGPT (mingpt.py:185) with YAML kwargs (+overrides), hash weights."""
    setup()
    from shapeformer.models.shapeformer.transformer.mingpt import CondTupleGPT
    y = load_yaml("configs/shapeformer/shapenet_scale.yaml")
    kw = dict(y["pl_model_opt"]["kwargs"]["transformer_opt"]["kwargs"])
    kw.update(overrides)
    m = CondTupleGPT(**kw)
    load_hash_weights(m)
    return m.eval()


def build_shapeformer(vq=None, **gpt_overrides):
    """Reference ShapeFormer with a seeded VQDIF in place of the missing ckpt."""
    setup()
    from shapeformer.models.shapeformer import representers
    from shapeformer.models.shapeformer.shapeformer import ShapeFormer
    vq = vq if vq is not None else build_vqdif(16)
    representers.Representer.init_trained_model_from_ckpt = lambda self, cfg: vq
    y = load_yaml("configs/shapeformer/shapenet_scale.yaml")
    kw = dict(y["pl_model_opt"]["kwargs"])
    kw["transformer_opt"] = dict(kw["transformer_opt"])
    kw["transformer_opt"]["kwargs"] = dict(kw["transformer_opt"]["kwargs"], **gpt_overrides)
    if "block_size" in gpt_overrides:
        kw["block_size"] = gpt_overrides["block_size"]
        kw["representer_opt"] = dict(kw["representer_opt"])
        kw["representer_opt"]["kwargs"] = dict(kw["representer_opt"]["kwargs"],
                                               block_size=gpt_overrides["block_size"])
    m = ShapeFormer(**kw)
    load_hash_weights(m.transformer)
    return m.eval()
