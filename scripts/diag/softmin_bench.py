"""Device time of one batched softmin launch (P = 64 pairs, N = M = 1024): python scripts/diag/softmin_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from livingscenes_amd.sinkhorn import _softmin_b
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
P, N, M = int(os.environ.get("P", "64")), 1024, 1024
x, y = torch.randn(P, N, 3, generator=g).to(dev) * 0.4, torch.randn(P, M, 3, generator=g).to(dev) * 0.4
pot = torch.randn(P, M, generator=g).to(dev) * 0.1
for epsv in (1.0, 0.0025):
    eps = torch.full((P,), epsv, device=dev)
    prev = torch.zeros(P, N, device=dev)
    for _ in range(5): _softmin_b(x, y, pot, -6.9, eps, prev=prev, average=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): _softmin_b(x, y, pot, -6.9, eps, prev=prev, average=True)
    e1.record(); torch.cuda.synchronize()
    print(f"variant {os.environ.get('LS_SOFTMIN_VARIANT', '0')} eps {epsv}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch")
