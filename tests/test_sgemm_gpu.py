"""GPU: csrc/sgemm.hip (plain f32 MFMA GEMM of the prefill / training step) against a float64 reference of the same product,
all operand-storage forms (y = x W^T, dX = dY W, dW = dY^T X), tails in M / N / K, accumulate and the fused epilogue."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dev, tA, tB, M, N, K, accumulate=False, bias=False, act=0, resid=False, seed=0, use_ws=True):
    from shapeformer_amd import _lib as L
    rs = np.random.RandomState(seed)
    A = rs.randn(*((K, M) if tA else (M, K))).astype(np.float32)
    B = rs.randn(*((N, K) if tB else (K, N))).astype(np.float32)
    C0 = rs.randn(M, N).astype(np.float32)
    bv, rv = rs.randn(N).astype(np.float32), rs.randn(M, N).astype(np.float32)
    dA, dB, dC = (torch.from_numpy(x).to(dev) for x in (A, B, C0.copy()))
    db, dr = torch.from_numpy(bv).to(dev), torch.from_numpy(rv).to(dev)
    ws = torch.empty(8 * M * N if use_ws else 4, device=dev)
    L.check(L.lib().sfmi_sgemm_mfma_f32(int(tA), int(tB), M, N, K, L.ptr(dA), A.shape[1], L.ptr(dB), B.shape[1], L.ptr(dC), N, int(accumulate),
                                        L.ptr(db) if bias else None, act, L.ptr(dr) if resid else None, L.ptr(ws) if use_ws else None, ws.numel(), 0.0, 0, L.stream_ptr()), "sgemm_mfma")
    opA = A.T.astype(np.float64) if tA else A.astype(np.float64)
    opB = B.T.astype(np.float64) if tB else B.astype(np.float64)
    ref = opA @ opB
    if accumulate:
        ref = ref + C0
    if bias:
        ref = ref + bv
    if act == 1:
        ref = np.maximum(ref, 0)
    elif act == 2:
        from scipy.special import erf
        ref = 0.5 * ref * (1 + erf(ref / np.sqrt(2)))
    if resid:
        ref = ref + rv
    got = dC.cpu().numpy().astype(np.float64)
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.mark.parametrize("tA,tB", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_all_operand_forms_with_tails(dev, tA, tB):
    for (M, N, K) in [(256, 256, 64), (300, 132, 100), (128, 4128, 1024), (1000, 384, 520), (4, 4, 4)]:
        if tA and M % 4:
            continue
        e = _run(dev, tA, tB, M, N, K, seed=M + N)
        assert e < 2e-6, (tA, tB, M, N, K, e)


def test_fused_epilogue_and_accumulate(dev):
    assert _run(dev, 0, 1, 515, 1024, 256, bias=True, act=2, resid=True) < 2e-6       # fc1-like: bias + GELU (+ residual)
    assert _run(dev, 0, 1, 515, 256, 1024, bias=True, act=1) < 2e-6
    from shapeformer_amd import _lib as L
    assert L.lib().sfmi_sgemm_mfma_splits(1024, 256, 3992) > 1                         # few output tiles, deep K: split-K path
    assert _run(dev, 1, 0, 1024, 256, 3992, accumulate=True) < 4e-6                   # dW += dY^T X (K = rows of the batch)
    assert _run(dev, 1, 0, 1024, 256, 3992, accumulate=True, use_ws=False) < 4e-6     # same without scratch (one slice)
    assert _run(dev, 0, 1, 256, 256, 4096, bias=True, act=2, resid=True) < 4e-6       # split-K with the full epilogue
    assert _run(dev, 0, 0, 3992, 1024, 512, accumulate=False) < 2e-6                  # dX = dY W


def test_stream_ptr_is_torchs_current_stream(dev):
    """_lib.stream_ptr() (one call per kernel launch) takes the raw-stream binding instead of building a Stream object: it must name
    the same hipStream_t as torch.cuda.current_stream(), also inside a stream context and after leaving it."""
    from shapeformer_amd import _lib as L
    assert L.stream_ptr() == torch.cuda.current_stream().cuda_stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert L.stream_ptr() == s.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert L.stream_ptr() == torch.cuda.current_stream().cuda_stream != s.cuda_stream
