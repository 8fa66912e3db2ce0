"""Time of the wide f16-split GEMM at the decoder shapes (dev tool; LS_LIB_PATH selects a timing-variant library, LS_GEMM_W2_PP the loop form)."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from livingscenes_amd import ops
dev = torch.device("cuda:0")
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = []
for (M, N, K) in [(262144, 768, 768), (65536, 768, 768), (6144, 5120, 256)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(dev); W = torch.randn(N, K, generator=g).to(dev); b = torch.randn(N, generator=g).to(dev)
    am, wm = ops.rowmax(A), ops.rowmax(W)
    planes = ops.presplit_w(W, wm) if K >= 512 else None
    t = min(timed(lambda: ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm, w_planes=planes)) for _ in range(3))
    msg = f"{M}x{N}x{K}: {t:7.1f} us"
    if N == K:   # the decoder's chain: the A operand's row maxima arrive as the previous layer's per-wave parts [M, 2 N / 128]
        o1, rm = ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm, w_planes=planes)
        t2 = min(timed(lambda: ops.gemm_chain(o1, W, b, relu=True, a_rowmax=rm, w_rowmax=wm, w_planes=planes)) for _ in range(3))
        msg += f" (chained, {rm.shape[1]} parts: {t2:7.1f})"
    out.append(msg)
print(os.environ.get("LS_TAG", ""), " | ".join(out), flush=True)
