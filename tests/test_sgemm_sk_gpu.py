"""GPU: csrc/sgemm_sk.hip (work-balanced f32 MFMA GEMM of the small-batch training step) against a float64 reference of the same
product: all operand-storage forms, both workgroup tiles, every grid size, tails in M / N / K, tiles cut between workgroups
(in-launch slab + ticket reduction), the fused epilogues (bias, GELU with the pre-activation side output, GELU' multiply, dropout,
residual, accumulate), re-armed tickets across launches and run-to-run bit identity."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Scratch:
    def __init__(self, dev):
        from shapeformer_amd import _lib as L
        self.slab = torch.empty(L.lib().sfmi_sgemm_sk_slab_floats(), device=dev)
        self.cnt = torch.zeros(1 << 20, device=dev, dtype=torch.int32)


@pytest.fixture(scope="module")
def scratch(dev):
    return _Scratch(dev)


def _tune(**kw):
    from shapeformer_amd import _lib as L
    for k, v in kw.items():
        L.check(L.lib().sfmi_tune_set(k.encode(), int(v)), f"tune {k}")


def _gelu(x):
    from scipy.special import erf
    return 0.5 * x * (1 + erf(x / np.sqrt(2)))


def _gelu_grad(x):
    from scipy.special import erf
    return 0.5 * (1 + erf(x / np.sqrt(2))) + x * np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


def _run(dev, sc, tA, tB, M, N, K, accumulate=False, bias=False, act=0, resid=False, c2=False, seed=0, reps=1):
    """-> (max relative error of C, of C2 or None, the device result)"""
    from shapeformer_amd import _lib as L
    rs = np.random.RandomState(seed)
    A = rs.randn(*((K, M) if tA else (M, K))).astype(np.float32)
    B = rs.randn(*((N, K) if tB else (K, N))).astype(np.float32)
    C0 = rs.randn(M, N).astype(np.float32)
    bv, rv, av = rs.randn(N).astype(np.float32), rs.randn(M, N).astype(np.float32), rs.randn(M, N).astype(np.float32)
    dA, dB = (torch.from_numpy(x).to(dev) for x in (A, B))
    db, dr, da = (torch.from_numpy(x).to(dev) for x in (bv, rv, av))
    for _ in range(reps):
        dC = torch.from_numpy(C0.copy()).to(dev)
        dC2 = torch.full((M, N), float("nan"), device=dev) if c2 else None
        L.check(L.lib().sfmi_sgemm_sk_f32(int(tA), int(tB), M, N, K, L.ptr(dA), A.shape[1], L.ptr(dB), B.shape[1], L.ptr(dC), L.ptr(dC2), N,
                                          int(accumulate), L.ptr(db) if bias else None, act, L.ptr(da) if act == 3 else None,
                                          L.ptr(dr) if resid else None, 0.0, 0, L.ptr(sc.slab), sc.slab.numel(), L.ptr(sc.cnt), sc.cnt.numel(),
                                          L.stream_ptr()), "sgemm_sk")
    opA = A.T.astype(np.float64) if tA else A.astype(np.float64)
    opB = B.T.astype(np.float64) if tB else B.astype(np.float64)
    ref = opA @ opB
    if accumulate:
        ref = ref + C0
    if bias:
        ref = ref + bv
    pre = ref
    if act == 1:
        ref = np.maximum(ref, 0)
    elif act == 2:
        ref = _gelu(ref)
    elif act == 3:
        ref = ref * _gelu_grad(av.astype(np.float64))
    if resid:
        ref = ref + rv
    got = dC.cpu().numpy().astype(np.float64)
    e = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))
    e2 = None
    if c2:
        e2 = float(np.abs(dC2.cpu().numpy().astype(np.float64) - pre).max() / (np.abs(pre).max() + 1e-30))
    return e, e2, dC


@pytest.mark.parametrize("tA,tB", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("tile", [1, 2])
def test_all_operand_forms_tiles_and_tails(dev, scratch, tA, tB, tile):
    _tune(sk_tile=tile, sk_grid=512)
    try:
        # (4,4,4): one unit; (300,132,100): ragged everything; (500,1024,1024): the step's proj shape (every tile cut);
        # (1024,260,499): a weight-gradient shape with K = tokens not a multiple of the chunk; (128,4128,1024): the padded heads
        for (M, N, K) in [(4, 4, 4), (256, 256, 64), (300, 132, 100), (500, 1024, 1024), (1024, 260, 499), (128, 4128, 1024), (1000, 384, 520)]:
            if (tA and M % 4) or ((not tA or tB) and K % 4):
                continue
            e, _, _ = _run(dev, scratch, tA, tB, M, N, K, seed=M + N + K)
            assert e < 2e-6, (tA, tB, tile, M, N, K, e)
    finally:
        _tune(sk_tile=0, sk_grid=512)
    assert int(scratch.cnt.abs().sum().item()) == 0, "a ticket counter was left armed"


@pytest.mark.parametrize("grid", [256, 512, 768, 1024])
def test_every_grid_size_cuts_tiles_correctly(dev, scratch, grid):
    _tune(sk_grid=grid, sk_tile=0)
    try:
        for (tA, tB, M, N, K) in [(0, 1, 499, 1024, 4096), (0, 0, 499, 1024, 3072), (1, 0, 3072, 1024, 499), (0, 1, 499, 4096, 1024), (0, 1, 3992, 1024, 1024),
                                  (1, 0, 4096, 1024, 499)]:
            e, _, _ = _run(dev, scratch, tA, tB, M, N, K, seed=grid + M)
            assert e < 3e-6, (grid, tA, tB, M, N, K, e)
    finally:
        _tune(sk_grid=512)
    assert int(scratch.cnt.abs().sum().item()) == 0


def test_fused_epilogues_accumulate_and_determinism(dev, scratch):
    _tune(sk_grid=512, sk_tile=0)
    e, e2, _ = _run(dev, scratch, 0, 1, 499, 4096, 1024, bias=True, act=2, c2=True)                    # fc1: h = GELU(pre), pre kept
    assert e < 2e-6 and e2 < 2e-6, (e, e2)
    e, _, _ = _run(dev, scratch, 0, 0, 499, 4096, 1024, act=3)                                         # dX of fc2 -> dhpre
    assert e < 2e-6, e
    e, _, _ = _run(dev, scratch, 0, 1, 499, 1024, 4096, bias=True, resid=True)                         # fc2 (every tile cut in 4)
    assert e < 3e-6, e
    e, _, _ = _run(dev, scratch, 0, 1, 515, 256, 1024, bias=True, act=1)
    assert e < 2e-6, e
    e, _, _ = _run(dev, scratch, 1, 0, 1024, 256, 3992, accumulate=True)                               # dW += dY^T X
    assert e < 4e-6, e
    e, e2, _ = _run(dev, scratch, 0, 1, 300, 516, 2048, bias=True, act=2, resid=True, c2=True, accumulate=True)
    assert e < 4e-6 and e2 < 4e-6, (e, e2)
    # the slices of a cut tile are added in k order whichever wave arrives last: launches are bit-identical
    outs = [_run(dev, scratch, 0, 1, 499, 1024, 4096, bias=True, seed=3, reps=3)[2] for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert int(scratch.cnt.abs().sum().item()) == 0


def test_dropout_mask_matches_the_plain_gemm(dev, scratch):
    """The dropout multiplier is the counter-hash of (seed, m * N + n): csrc/sgemm.hip and csrc/sgemm_sk.hip drop the same elements."""
    from shapeformer_amd import _lib as L
    M, N, K = 499, 1024, 1024
    x, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    y0, y1 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    L.check(L.lib().sfmi_sgemm_mfma_f32(0, 1, M, N, K, L.ptr(x), K, L.ptr(W), K, L.ptr(y0), N, 0, None, 0, None, None, 0, 0.25, 77, L.stream_ptr()), "sgemm")
    L.check(L.lib().sfmi_sgemm_sk_f32(0, 1, M, N, K, L.ptr(x), K, L.ptr(W), K, L.ptr(y1), None, N, 0, None, 0, None, None, 0.25, 77, L.ptr(scratch.slab),
                                      scratch.slab.numel(), L.ptr(scratch.cnt), scratch.cnt.numel(), L.stream_ptr()), "sgemm_sk")
    assert torch.equal(y0 == 0, y1 == 0) and 0.2 < float((y1 == 0).float().mean()) < 0.3
    assert float((y0 - y1).abs().max()) < 2e-4 * float(y0.abs().max())


def test_argument_checks(dev, scratch):
    from shapeformer_amd import _lib as L
    x = torch.zeros(256, 256, device=dev)
    small = torch.zeros(8, device=dev, dtype=torch.int32)
    assert L.lib().sfmi_sgemm_sk_cnt_ints(256, 256) == 64

    def call(cnt, act=0, aux=None):
        return L.lib().sfmi_sgemm_sk_f32(0, 1, 256, 256, 256, L.ptr(x), 256, L.ptr(x), 256, L.ptr(x), None, 256, 0, None, act, aux, None, 0.0, 0,
                                         L.ptr(scratch.slab), scratch.slab.numel(), L.ptr(cnt), cnt.numel(), L.stream_ptr())
    assert call(small) == -1                  # too few ticket counters for 16 tiles
    assert call(scratch.cnt, act=3) == -1     # act 3 needs aux
    assert call(scratch.cnt) == 0
