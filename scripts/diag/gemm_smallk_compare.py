"""K = 32 / 64 table GEMM shapes: persistent f16-split kernel (default) vs the tiled kernel (LS_GEMM_PERSIST=0): bit-identity + timing."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from livingscenes_amd import ops
dev = torch.device("cuda:0")
for (name, M, N, K) in (("L1 table", 196608, 128, 32), ("L2 P", 196608, 256, 32), ("L3 P", 98304, 256, 64), ("L4 P", 98304, 512, 64), ("ragged", 70001, 200, 64), ("ragged32", 33333, 130, 32)):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.2).to(dev)
    out = ops.gemm(A, W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(A, W)
    e1.record(); torch.cuda.synchronize()
    ref = A.double() @ W.double().T
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    print(M, N, K, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], f"{e0.elapsed_time(e1) / 10 * 1e3:.1f}us", f"err {err:.2e}", f"write {M*N*4/(e0.elapsed_time(e1)/10)/1e9:.2f}TB/s", name)
''' % ROOT
res = {}
for name, env in (("persistent", {}), ("tiled", {"LS_GEMM_PERSIST": "0"})):
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l and l[0].isdigit()]
    res[name] = lines
    print("==", name); print("\n".join(lines))
    if out.returncode: print(out.stderr[-2000:])
ok = [l.split()[3] for l in res["persistent"]] == [l.split()[3] for l in res["tiled"]]
print("bit-identical:", ok)
sys.exit(0 if ok else 1)
