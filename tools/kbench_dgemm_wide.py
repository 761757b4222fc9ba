"""Wide decode-GEMM (64 < M <= 192 rows, csrc/gpt.hip dgemm_wide_kernel) check + timing: graph of 24 layer launches.
Env: B (rows), WIDE=1 forces the wide kernel for B <= 64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shapeformer_amd import _lib as L
from shapeformer_amd.gpt import CondTupleGPT
from bench import ev_time
dev = torch.device("cuda:0")
gpt = CondTupleGPT(device=dev)
D = gpt.D
lib = L.lib()


def pack(x, Mp):   # (M,N) row-major -> fragment-packed [Mp/16][N/16][64][4]
    M, N = x.shape
    xp = torch.zeros(Mp, N, device=x.device); xp[:M] = x
    return xp.view(Mp // 16, 16, N // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous()   # [mt][nt][q][ml][j]


def unpack(p, M, N):
    Mp = p.numel() // N
    return p.view(Mp // 16, N // 16, 4, 16, 4).permute(0, 3, 1, 2, 4).reshape(Mp, N)[:M]


for B in [int(b) for b in os.environ.get("B", "64,192,256").split(",")]:
    st = gpt._alloc(B, 8)
    Bp = st["resid"].shape[0]
    fn = lib.sfmi_decode_gemm_wide_f32 if (B > 64 or os.environ.get("WIDE")) else lib.sfmi_decode_gemm_f32
    ly = gpt.layers[0]
    torch.manual_seed(0)
    for nm, wp, c1, c2, N, K, ln, act, use_res, Ss in (
            ("qkv", "pqkv", "c1qkv", "c2qkv", 3 * D, D, 1, 0, False, (1, 2)),
            ("proj", "pproj", None, "bproj", D, D, 0, 0, True, (2, 4, 8)),
            ("fc1", "pfc1", "c1fc1", "c2fc1", 4 * D, D, 1, 1, False, (1, 2)),
            ("fc2", "pfc2", None, "bfc2", D, 4 * D, 0, 0, True, (4, 8))):
        x = torch.randn(B, K, device=dev)
        xp = pack(x, Bp)
        res = torch.randn(B, N, device=dev) if use_res else None
        resp = pack(res, Bp) if use_res else None
        out = torch.zeros(Bp * N, device=dev)
        ref = None
        for S in Ss:
            if (K // S) % 128:
                continue
            def one(l):
                L.check(fn(L.ptr(xp), L.ptr(getattr(l, wp)), L.ptr(getattr(l, c1)) if c1 else None, L.ptr(getattr(l, c2)),
                           L.ptr(resp), L.ptr(out), B, N, K, N, ln, act, 1, S, L.ptr(st["slab"]), L.ptr(st["cnt"]),
                           L.stream_ptr()), "dgemm")
            one(ly); torch.cuda.synchronize()
            got = unpack(out, B, N).clone()
            if ref is None:
                ref = got   # first S is the comparison base; checked against the narrow kernel below
                if B <= 64:
                    o2 = torch.zeros_like(out)
                    L.check(lib.sfmi_decode_gemm_f32(L.ptr(xp), L.ptr(getattr(ly, wp)), L.ptr(getattr(ly, c1)) if c1 else None,
                                                     L.ptr(getattr(ly, c2)), L.ptr(resp), L.ptr(o2), B, N, K, N, ln, act, 1, 1,
                                                     None, None, L.stream_ptr()), "narrow")
                    torch.cuda.synchronize()
                    print(f"   vs narrow kernel: max|d| {float((unpack(o2, B, N) - got).abs().max()):.3e}")
                else:   # rows 0..63 through the narrow kernel
                    x64 = pack(x[:64], 64); r64 = pack(res[:64], 64) if use_res else None
                    o2 = torch.zeros(64 * N, device=dev)
                    L.check(lib.sfmi_decode_gemm_f32(L.ptr(x64), L.ptr(getattr(ly, wp)), L.ptr(getattr(ly, c1)) if c1 else None,
                                                     L.ptr(getattr(ly, c2)), L.ptr(r64), L.ptr(o2), 64, N, K, N, ln, act, 1, 1,
                                                     None, None, L.stream_ptr()), "narrow")
                    torch.cuda.synchronize()
                    print(f"   vs narrow kernel (rows 0..63): max|d| {float((unpack(o2, 64, N) - got[:64]).abs().max()):.3e}"
                          f"  scale {float(got.abs().max()):.2f}")
            else:
                print(f"   S={S} vs first: max|d| {float((got - ref).abs().max()):.3e}")
            def body():
                for l in gpt.layers:
                    one(l)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body()
            ms = ev_time(g.replay, 10) / len(gpt.layers)
            print(f"B={B} {nm:5s} S={S:2d} {ms*1e3:7.2f} us  {2*B*N*K/ms/1e9:7.1f} TFLOP/s  W {N*K*4/ms/1e6:7.1f} GB/s")
