"""fp32-MFMA GEMM micro-benchmark at the encoder's table-GEMM shapes (out[M,N] = A[M,K] W[N,K]^T)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import ops
dev = torch.device("cuda:0")
shapes = [("L1 table", 196608, 128, 32), ("L2 P", 196608, 256, 32), ("L2 Q", 98304, 384, 32), ("L3 table", 98304, 640, 64),
          ("L4 P", 98304, 512, 64), ("L4 Q", 24576, 768, 64), ("L5 P", 24576, 1024, 128), ("L5 Q", 6144, 1536, 128),
          ("L6 table", 6144, 5120, 256), ("glob6", 6144, 1024, 512), ("mean-part 6", 192, 1024, 512), ("decoder", 262144, 768, 768)]
if "--check" in sys.argv:
    # accuracy against fp64: max |out - ref| in units of 2^-24 * (|A| |W|^T) (the fp32 FMA-chain bound is ~K/2 of these units)
    sys.argv.remove("--check")
    for K in (32, 64, 256, 768):
        g = torch.Generator(device="cpu").manual_seed(K)
        A = (torch.randn(4096, K, generator=g) * torch.exp(torch.randn(4096, K, generator=g))).to(dev)
        W = (torch.randn(512, K, generator=g) * torch.exp(torch.randn(512, K, generator=g))).to(dev)
        out = ops.gemm(A, W).double()
        ref = A.double() @ W.double().T
        bound = (A.double().abs() @ W.double().abs().T) * 2.0 ** -24
        print(f"K={K:4d}: max err / (2^-24 sum|a||w|) = {((out - ref).abs() / bound).max().item():.3f}   rel-to-max {((out - ref).abs().max() / ref.abs().max()).item():.2e}")
if len(sys.argv) > 1: shapes = [s for s in shapes if sys.argv[1] in s[0]]
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
    # three forms, interleaved twice (the chip's clock drifts under load): ls_gemm_f32 (the kernel scans its operand rows for the range
    # scaling), ls_gemm_f32_ex with the row maxima of A and W handed over, the same + writing the output's row maxima
    am, wm = ops.rowmax(A), ops.rowmax(W)
    forms = [lambda: ops.gemm(A, W), lambda: ops.gemm_chain(A, W, a_rowmax=am, w_rowmax=wm, want_rowmax=False),
             lambda: ops.gemm_chain(A, W, a_rowmax=am, w_rowmax=wm, want_rowmax=True)]
    planes = ops.presplit_w(W, wm)     # the weight's f16 pieces, split once (K >= 128)
    if planes is not None:
        forms.append(lambda: ops.gemm_chain(A, W, a_rowmax=am, w_rowmax=wm, want_rowmax=True, w_planes=planes))
        assert torch.equal(forms[2]()[0], forms[3]()[0]), "planes path differs"
    if ops.load().ls_gemm_workspace_bytes(M, N, K):   # split-K shape: the chained forms never split
        forms = forms[:1]
    ms = [min(timed(f), timed(f)) for f in forms]
    wr = M * N * 4 / ms[0] / 1e9; fl = 2.0 * M * N * K / ms[0] / 1e9
    extra = "" if len(ms) == 1 else f"   maxima given {ms[1]*1e3:8.1f} us   + emitted {ms[2]*1e3:8.1f} us"
    if len(ms) == 4:
        extra += f"   + W pre-split {ms[3]*1e3:8.1f} us"
    print(f"{name:12s} M={M:7d} N={N:5d} K={K:4d}: {ms[0]*1e3:8.1f} us   write {wr:6.2f} TB/s   {fl:6.1f} TFLOP/s{extra}")
