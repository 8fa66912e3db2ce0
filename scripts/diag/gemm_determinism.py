"""Diagnostic: is ls_gemm_f32 bit-reproducible when eight streams run it concurrently (same inputs)?"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(196608, 128, 32), (196608, 384, 32), (98304, 640, 64), (98304, 256, 64), (24576, 1280, 128), (6144, 2560, 256), (6144, 1024, 512), (65536, 768, 768)]
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
for (M, N, K) in shapes:
    A = (torch.randn(M, K, generator=g) * 0.3).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.1).to(dev)
    ref = ops.gemm(A, W)
    torch.cuda.synchronize()
    bad = 0
    worst = 0.0
    for rep in range(3):
        outs = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                outs.append(ops.gemm(A, W))
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                worst = max(worst, float((o - ref).abs().max() / ref.abs().max()))
    print((M, N, K), "differing outputs:", bad, "of 24, worst rel", worst)
