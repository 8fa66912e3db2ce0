"""gemm_w2_kernel (256 x 256 tiles) against gemm_h2_kernel (128 x 128): bit-identity and time per shape, the wide kernel with both waves of a SIMD in step
(LS_GEMM_W2_PP=0, the default) and half a step apart (=1), the pre-split W planes through registers (LS_GEMM_W2_DIRECT=0, the default) and by LDS-direct loads (=1).  The switches are read once per process, so each setting runs in a child."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from livingscenes_amd import ops
dev = torch.device("cuda:0")
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
import hashlib
for (M, N, K) in [(1000, 300, 128), (777, 520, 256), (4096, 768, 768), (6144, 1536, 128), (24576, 1024, 128), (6144, 5120, 256), (6144, 1024, 512),
                  (32768, 768, 768), (65536, 768, 768), (262144, 768, 768), (1024, 768, 768), (8192, 768, 768), (16384, 512, 512)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(dev); W = torch.randn(N, K, generator=g).to(dev); b = torch.randn(N, generator=g).to(dev)
    am, wm = ops.rowmax(A), ops.rowmax(W)
    out, rm = ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm)
    ref = (A.double() @ W.double().T + b.double()).clamp_min(0)
    bound = (A.double().abs() @ W.double().abs().T) * 2.0 ** -24
    err = ((out.double() - ref).abs() / bound).max().item()
    h = hashlib.md5(out.cpu().numpy().tobytes() + rm.cpu().numpy().tobytes()).hexdigest()[:10]
    t = min(timed(lambda: ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm)) for _ in range(2))
    planes = ops.presplit_w(W, wm)
    out2, rm2 = ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm, w_planes=planes)
    same = torch.equal(out, out2) and torch.equal(rm, rm2)
    t2 = min(timed(lambda: ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm, w_planes=planes)) for _ in range(2))
    print(f"{M:7d} {N:5d} {K:4d}  err {err:5.2f}  md5 {h}  {t:8.1f} us   W pre-split {t2:8.1f} us  (identical: {same})")
''' % ROOT
for w, pp, dw in (("0", "0", "1"), ("1", "0", "0"), ("1", "0", "1"), ("1", "1", "1")):
    print("LS_GEMM_WIDE =", w, " LS_GEMM_W2_PP =", pp, " LS_GEMM_W2_DIRECT =", dw, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LS_GEMM_WIDE=w, LS_GEMM_W2_PP=pp, LS_GEMM_W2_DIRECT=dw))
