"""Raw-cloud FPS (N points -> 1024): bucketed pruned scan vs LS_FPS_FULL_SCAN=1.  python scripts/diag/fps_raw_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from livingscenes_amd import ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for N, B in ((10000, 1), (25000, 1), (60000, 1), (60000, 32), (15000, 64)):
    u = rng.random((B, N, 2)).astype(np.float32)
    p = np.concatenate([u * np.float32(2.0), (0.2 * np.sin(4 * u[..., :1])).astype(np.float32)], -1)
    x = torch.from_numpy(p).to(dev)
    for _ in range(2):
        ops.fps(x, 1024)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        ops.fps(x, 1024)
    torch.cuda.synchronize()
    print(f"N={N} B={B}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms per call ({'full scan' if os.environ.get('LS_FPS_FULL_SCAN') else 'bucketed'})", flush=True)
