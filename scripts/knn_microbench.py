"""k-NN micro-benchmark at one encoder-layer shape (for rocprofv3 PMC passes and A/B timing)."""
import argparse, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=64); ap.add_argument("--Nd", type=int, default=1024)
ap.add_argument("--Ns", type=int, default=1024); ap.add_argument("--C", type=int, default=32)
ap.add_argument("--iters", type=int, default=5); ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
src = torch.randn(a.B, a.Ns, 3, a.C, generator=g).to(dev)
dst = src[:, :a.Nd].contiguous()
seeds = None
if a.flags & 8:  # bit 3: use the exact answer of a first run as hints (upper bound of what seeding can give)
    seeds = ops.knn(dst, src, 16)
    a.flags &= ~8
for _ in range(2): ops.knn(dst, src, 16, flags=a.flags, seeds=seeds)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters): ops.knn(dst, src, 16, flags=a.flags, seeds=seeds)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
fl = 3.0 * a.B * a.Nd * a.Ns * 3 * a.C
print(f"knn B={a.B} Nd={a.Nd} Ns={a.Ns} C={a.C} flags={a.flags}: {ms:.4f} ms/launch  {fl/ms/1e9:.2f} TFLOP/s (3 flops/pair-dim)")
