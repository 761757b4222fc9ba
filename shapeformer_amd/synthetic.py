"""Seeded synthetic point clouds with the reference's item schema.

The reference's datasets (IMNet2_64 HDF5, imnet_datasets.py:144-224) are not available; benchmarks and
parity tests use analytic surfaces instead: union of 1-3 primitives (sphere / torus / box), small
Gaussian noise, coordinates in [-1,1]; the partial "context" cloud keeps the points whose normal faces a
random view direction (a cheap stand-in for the virtual-scan selector, partial.py:127-146).
"""
from __future__ import annotations

import numpy as np


def _sphere(rs, n):
    v = rs.randn(n, 3)
    nrm = v / np.linalg.norm(v, axis=1, keepdims=True)
    r = rs.uniform(0.25, 0.5)
    return nrm * r, nrm


def _torus(rs, n):
    R, r = rs.uniform(0.3, 0.45), rs.uniform(0.08, 0.15)
    a, b = rs.uniform(0, 2 * np.pi, n), rs.uniform(0, 2 * np.pi, n)
    p = np.stack([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)], 1)
    nrm = np.stack([np.cos(b) * np.cos(a), np.cos(b) * np.sin(a), np.sin(b)], 1)
    return p, nrm


def _box(rs, n):
    h = rs.uniform(0.15, 0.45, 3)
    face = rs.randint(0, 6, n)
    p = rs.uniform(-1, 1, (n, 3)) * h
    nrm = np.zeros((n, 3))
    ax, sg = face // 2, (face % 2) * 2 - 1
    p[np.arange(n), ax] = sg * h[ax]
    nrm[np.arange(n), ax] = sg
    return p, nrm


def make_shape(seed, n_full=32768, n_partial=16384):
    """-> dict(Xbd (n_full,3), Xct (n_partial,3)) float32 in [-1,1]."""
    rs = np.random.RandomState(seed)
    k = rs.randint(1, 4)
    pts, nrms = [], []
    for _ in range(k):
        p, nr = [_sphere, _torus, _box][rs.randint(0, 3)](rs, n_full)
        q, _ = np.linalg.qr(rs.randn(3, 3))
        c = rs.uniform(-0.25, 0.25, 3)
        pts.append(p @ q.T + c)
        nrms.append(nr @ q.T)
    P, N = np.concatenate(pts), np.concatenate(nrms)
    sel = rs.choice(len(P), n_full, replace=False)
    P, N = P[sel], N[sel]
    P = np.clip(P + rs.randn(*P.shape) * 0.005, -0.95, 0.95)
    v = rs.randn(3)
    v /= np.linalg.norm(v)
    vis = np.nonzero(N @ v > 0)[0]
    if len(vis) < 16:
        vis = np.arange(len(P))
    Xct = P[rs.choice(vis, n_partial, replace=True)]
    return dict(Xbd=P.astype(np.float32), Xct=Xct.astype(np.float32))


def make_batch(seed0, B, n_full=32768, n_partial=16384):
    items = [make_shape(seed0 + i, n_full, n_partial) for i in range(B)]
    return {k: np.stack([it[k] for it in items]) for k in ("Xbd", "Xct")}
