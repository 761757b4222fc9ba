"""Deterministic, RNG-free weight generator keyed by state-dict key name.

No checkpoints ship with the reference (`experiments/` absent), so
parity and benchmarks run on *generated* weights.  The generator is a pure
counter hash (no torch / numpy RNG): both this container (reference import,
oracle) and the GPU box rebuild bit-identical full-size weights from the key
name + shape alone, so no weight files ever need to travel.

Key names / shapes are those of the reference state dicts
(SURVEY.md §8(b) B2): `encoder.fc_pos.weight`, `decoder.unet3d.encoders.0...`,
`quantizer.embedding.weight`, `blocks.0.3.attn.key.weight`, ...
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M32 = np.uint32(0xFFFFFFFF)


def _fnv1a32(s: str) -> int:
    h = 0x811C9DC5
    for ch in s.encode("utf-8"):
        h ^= ch
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


def hash_unit(key: str, n: int) -> np.ndarray:
    """n floats in [0,1), element i = murmur3-finalizer(i * golden + fnv(key))."""
    seed = np.uint32(_fnv1a32(key))
    with np.errstate(over="ignore"):
        h = np.arange(n, dtype=np.uint32)
        h *= np.uint32(0x9E3779B1)
        h += seed
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))


def hash_uniform(key: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_unit(key, n)
    out = (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)
    return out.reshape(shape)


def _kind(key: str, shape) -> str:
    leaf = key.split(".")[-1]
    if key.endswith("attn.mask"):
        return "tril"
    if key == "quantizer.N" or key.endswith("quantizer.N"):
        return "zeros"
    is_norm = (".groupnorm." in key or ".ln1." in key or ".ln2." in key
               or (".heads." in ("." + key) and key.split(".")[-2] == "0"))
    if is_norm:
        return "norm_w" if leaf == "weight" else "norm_b"
    if leaf == "bias":
        return "bias"
    if ("embedding" in key or "tok_embs" in key or key.endswith("pos_emb")
            or key.endswith("z_avg")):
        return "emb"
    if ".heads." in ("." + key):
        return "head"
    return "matrix"


def make_tensor(key: str, shape) -> np.ndarray:
    """Generate the tensor for state-dict entry `key` of shape `shape` (float32)."""
    shape = tuple(int(s) for s in shape)
    kind = _kind(key, shape)
    if kind == "tril":
        n = shape[-1]
        return np.tril(np.ones((n, n), np.float32)).reshape(shape)
    if kind == "zeros":
        return np.zeros(shape, np.float32)
    if kind == "norm_w":
        return hash_uniform(key, shape, 0.9, 1.1)
    if kind in ("norm_b", "bias"):
        return hash_uniform(key, shape, -0.1, 0.1)
    if kind == "emb":
        # z_avg mirrors embedding.weight in the reference ctor (quantizer.py:18)
        k = key.replace("z_avg", "embedding.weight")
        return hash_uniform(k, shape, -1.0, 1.0)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    s = math.sqrt(3.0 / fan_in)
    if kind == "head":
        s *= 4.0
    return hash_uniform(key, shape, -s, s)


# --------------------------------------------------------------------------
# state-dict layouts (names follow the reference modules)
# --------------------------------------------------------------------------

def vqdif_spec(res: int = 16) -> "OrderedDict[str, tuple]":
    """(key -> shape) of the reference VQDIF state dict for the shipped YAMLs.

    res=16: configs/vqdif/shapenet_res16.yaml (d=128, 2 down/up steps)
    res=32: configs/vqdif/shapenet_res32.yaml (d=64, 1 down/up step)
    """
    assert res in (16, 32)
    hd, cd = 32, 32
    steps = 2 if res == 16 else 1
    d = 32 * (2 ** steps)
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    # encoder (enc.py:34-40)
    sp["encoder.fc_pos.weight"] = (2 * hd, 3)
    sp["encoder.fc_pos.bias"] = (2 * hd,)
    for i in range(5):
        p = f"encoder.blocks.{i}."
        sp[p + "fc_0.weight"] = (hd, 2 * hd)
        sp[p + "fc_0.bias"] = (hd,)
        sp[p + "fc_1.weight"] = (hd, hd)
        sp[p + "fc_1.bias"] = (hd,)
        sp[p + "shortcut.weight"] = (hd, 2 * hd)
    sp["encoder.fc_c.weight"] = (cd, hd)
    sp["encoder.fc_c.bias"] = (cd,)
    c = cd
    for s in range(steps):
        p = f"encoder.downsampler.blocks.{2 * s}."
        sp[p + "conv.weight"] = (2 * c, c, 2, 2, 2)
        sp[p + "groupnorm.weight"] = (2 * c,)
        sp[p + "groupnorm.bias"] = (2 * c,)
        p = f"encoder.downsampler.blocks.{2 * s + 1}."
        sp[p + "conv.weight"] = (2 * c, 2 * c, 1, 1, 1)
        sp[p + "groupnorm.weight"] = (2 * c,)
        sp[p + "groupnorm.bias"] = (2 * c,)
        c *= 2
    assert c == d
    # decoder unet3d (unet3d.py:361-474, 'gcr', f_maps d,2d,4d)
    f = [d, 2 * d, 4 * d]

    def single(prefix, cin, cout):
        sp[prefix + "groupnorm.weight"] = (cin,)
        sp[prefix + "groupnorm.bias"] = (cin,)
        sp[prefix + "conv.weight"] = (cout, cin, 3, 3, 3)

    def double(prefix, cin, cout, encoder):
        if encoder:
            c1 = max(cout // 2, cin)
            single(prefix + "SingleConv1.", cin, c1)
            single(prefix + "SingleConv2.", c1, cout)
        else:
            single(prefix + "SingleConv1.", cin, cout)
            single(prefix + "SingleConv2.", cout, cout)

    double("decoder.unet3d.encoders.0.basic_module.", d, f[0], True)
    double("decoder.unet3d.encoders.1.basic_module.", f[0], f[1], True)
    double("decoder.unet3d.encoders.2.basic_module.", f[1], f[2], True)
    double("decoder.unet3d.decoders.0.basic_module.", f[2] + f[1], f[1], False)
    double("decoder.unet3d.decoders.1.basic_module.", f[1] + f[0], f[0], False)
    sp["decoder.unet3d.final_conv.weight"] = (d, f[0], 1, 1, 1)
    sp["decoder.unet3d.final_conv.bias"] = (d,)
    c = d
    for s in range(steps):
        for j, (ci, co) in enumerate(((c, c // 2), (c // 2, c // 2))):
            p = f"decoder.upsampler.blocks.{3 * s + 1 + j}."
            sp[p + "conv.weight"] = (co, ci, 3, 3, 3)
            sp[p + "groupnorm.weight"] = (co,)
            sp[p + "groupnorm.bias"] = (co,)
        c //= 2
    for i in range(5):
        sp[f"decoder.fc_c.{i}.weight"] = (hd, cd)
        sp[f"decoder.fc_c.{i}.bias"] = (hd,)
    sp["decoder.fc_p.weight"] = (hd, 3)
    sp["decoder.fc_p.bias"] = (hd,)
    for i in range(5):
        p = f"decoder.blocks.{i}."
        sp[p + "fc_0.weight"] = (hd, hd)
        sp[p + "fc_0.bias"] = (hd,)
        sp[p + "fc_1.weight"] = (hd, hd)
        sp[p + "fc_1.bias"] = (hd,)
    sp["decoder.fc_out.weight"] = (1, hd)
    sp["decoder.fc_out.bias"] = (1,)
    sp["quantizer.embedding.weight"] = (4096, d)
    sp["quantizer.N"] = (4096,)
    sp["quantizer.z_avg"] = (4096, d)
    return sp


def gpt_spec(n_embd=1024, n_layers=(20, 4), block_size=812, vocab_sizes=(4097, 4097),
             extra_vocab_sizes=(4097,), with_masks: bool = False) -> "OrderedDict[str, tuple]":
    """(key -> shape) of the reference CondTupleGPT state dict (mingpt.py:185-254).

    `with_masks` adds the 24 persistent (1,1,bs,bs) causal-mask buffers the
    reference registers (mingpt.py:71); the build never materialises them.
    """
    D = n_embd
    sp: "OrderedDict[str, tuple]" = OrderedDict()
    sp["pos_emb"] = (1, block_size, D)
    sp["cond_pos_emb"] = (1, block_size, D)
    for i, v in enumerate(vocab_sizes):
        sp[f"tok_embs.{i}.weight"] = (v, D)
    for i, v in enumerate(extra_vocab_sizes):
        sp[f"extra_tok_embs.{i}.weight"] = (v, D)
    for s, nl in enumerate(n_layers):
        for n in range(nl):
            p = f"blocks.{s}.{n}."
            for ln in ("ln1", "ln2"):
                sp[p + ln + ".weight"] = (D,)
                sp[p + ln + ".bias"] = (D,)
            for m in ("key", "query", "value", "proj"):
                sp[p + f"attn.{m}.weight"] = (D, D)
                sp[p + f"attn.{m}.bias"] = (D,)
            if with_masks:
                sp[p + "attn.mask"] = (1, 1, block_size, block_size)
            sp[p + "mlp.0.weight"] = (4 * D, D)
            sp[p + "mlp.0.bias"] = (4 * D,)
            sp[p + "mlp.2.weight"] = (D, 4 * D)
            sp[p + "mlp.2.bias"] = (D,)
    for s, v in enumerate(vocab_sizes):
        sp[f"heads.{s}.0.weight"] = (D,)
        sp[f"heads.{s}.0.bias"] = (D,)
        sp[f"heads.{s}.1.weight"] = (v, D)
    return sp


def make_state_dict(spec, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Generate every tensor of `spec`; the hash key is the un-prefixed name."""
    return OrderedDict((prefix + k, make_tensor(k, shp)) for k, shp in spec.items())


# --------------------------------------------------------------------------
# same generator on a torch device (bit-identical to the numpy path; used so the
# 325 M-parameter transformer can be materialised on the GPU in milliseconds)
# --------------------------------------------------------------------------
def hash_unit_torch(key: str, n: int, device):
    import torch
    M = 0xFFFFFFFF
    h = torch.arange(n, dtype=torch.int64, device=device)
    h = (h * 0x9E3779B1 + _fnv1a32(key)) & M
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M
    h = h ^ (h >> 16)
    return (h >> 8).to(torch.float32) * (1.0 / (1 << 24))


def make_tensor_torch(key: str, shape, device):
    import torch
    shape = tuple(int(s) for s in shape)
    kind = _kind(key, shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if kind in ("tril", "zeros"):
        return torch.from_numpy(make_tensor(key, shape)).to(device)

    def uni(k, lo, hi):
        u = hash_unit_torch(k, n, device)
        lo32, d32 = np.float32(lo), np.float32(hi - lo)
        return (float(lo32) + u * float(d32)).reshape(shape)

    if kind == "norm_w":
        return uni(key, 0.9, 1.1)
    if kind in ("norm_b", "bias"):
        return uni(key, -0.1, 0.1)
    if kind == "emb":
        return uni(key.replace("z_avg", "embedding.weight"), -1.0, 1.0)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    s = math.sqrt(3.0 / fan_in)
    if kind == "head":
        s *= 4.0
    return uni(key, -s, s)
