"""The GEMM launch forms must agree BIT FOR BIT (same products, same accumulation order): the pipelined kernel (default for K >= 128)
and the two-barrier kernel (LS_GEMM_H2_SIMPLE=1).  Also prints timings and the error against fp64.
python scripts/diag/gemm_variants_compare.py"""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, hashlib, torch
sys.path.insert(0, %r)
from livingscenes_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in ((262144, 512, 512), (131072, 512, 512), (65536, 512, 256), (70001, 500, 136), (262144, 768, 768), (33000, 1024, 128)):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev); W = (torch.randn(N, K, generator=g) * 0.05).to(dev); b = torch.randn(N, generator=g).to(dev)
    out = ops.gemm(A, W, bias=b, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(A, W, bias=b, relu=True)
    e1.record(); torch.cuda.synchronize()
    ref = torch.relu(A.double() @ W.double().T + b.double())
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    print(M, N, K, hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16], f"{e0.elapsed_time(e1) / 5 * 1e3:.1f}us", f"err {err:.2e}", f"{2.0*M*N*K/(e0.elapsed_time(e1)/5)/1e9:.0f}TF")
''' % ROOT
res = {}
for name, env in (("pipelined", {}), ("two-barrier", {"LS_GEMM_H2_SIMPLE": "1"})):
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if l and l[0].isdigit()]
    res[name] = lines
    print("==", name); print("\n".join(lines)); 
    if out.returncode: print(out.stderr[-2000:])
keys = list(res)
ok = all([l.split()[3] for l in res[keys[0]]] == [l.split()[3] for l in res[k]] for k in keys[1:])
print("bit-identical across launch forms:", ok)
sys.exit(0 if ok else 1)
