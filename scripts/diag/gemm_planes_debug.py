import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from livingscenes_amd import ops
dev = torch.device("cuda:0")
for (M, N, K, scaled) in [(640, 768, 768, False), (640, 768, 768, True), (256, 128, 128, True)]:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = torch.randn(N, K, generator=g)
    if scaled: W = W * torch.exp2(torch.randint(-30, 30, (N, 1), generator=g).float())
    W = W.to(dev)
    am, wm = ops.rowmax(A), ops.rowmax(W)
    planes = ops.presplit_w(W, wm)
    o0, r0 = ops.gemm_chain(A, W, None, a_rowmax=am, w_rowmax=wm)
    o1, r1 = ops.gemm_chain(A, W, None, a_rowmax=am, w_rowmax=wm, w_planes=planes)
    d = (o0 != o1)
    print(M, N, K, scaled, "differ:", int(d.sum()), "cols differing:", d.any(0).nonzero().flatten()[:20].tolist(), "rows:", d.any(1).nonzero().flatten()[:10].tolist())
    if d.any():
        c = int(d.any(0).nonzero()[0]); print(" col", c, "o0", o0[:3, c].tolist(), "o1", o1[:3, c].tolist(), "ratio", (o1[:3, c] / o0[:3, c]).tolist(), "log2 wmax", torch.log2(wm[c]).item())
