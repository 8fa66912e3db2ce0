"""Ragged farthest-point sampling front-end (Shape_Prior.encode_fps): B raw instance clouds of up to P points
-> 1024 points each in one launch, then one batched encode."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B, P in ((24, 60000), (24, 8192), (64, 4096), (8, 60000)):
    pts = torch.randn(B, P, 3, generator=g).to(dev)
    lengths = torch.randint(P // 2, P + 1, (B,), generator=g).to(dev)
    for _ in range(2): ops.fps(pts, 1024, lengths=lengths)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): idx = ops.fps(pts, 1024, lengths=lengths)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"ragged FPS B={B} P<={P} -> 1024: {dt*1e3:.2f} ms ({dt/1024*1e6:.2f} us per selection step)")
