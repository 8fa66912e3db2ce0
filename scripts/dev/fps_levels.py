"""FPS at the encoder's three down-sampling levels (1024 -> 512, 512 -> 128, 128 -> 32; 64 instances), alone on the device: us per launch and
ns per dependent arg-max step.  python scripts/dev/fps_levels.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from livingscenes_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B, N, K in ((64, 1024, 512), (64, 512, 128), (64, 128, 32), (64, 2048, 1024), (64, 256, 128)):
    pts = torch.randn(B, N, 3, generator=g).to(dev)
    for _ in range(3): ops.fps(pts, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): idx = ops.fps(pts, K)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"FPS B={B} N={N} -> {K}: {dt*1e6:.1f} us per launch, {dt/K*1e9:.0f} ns per step")
