#!/bin/bash
# shader clock and socket power while the wide GEMM runs back to back (dev tool): product library with both loop forms, then the variants given
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # tag, env...
  tag=$1; shift
  env "$@" python - <<'PY' 2>/dev/null &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from livingscenes_amd import ops
dev = torch.device("cuda:0")
M, N, K = 262144, 768, 768
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).to(dev); W = torch.randn(N, K, generator=g).to(dev); b = torch.randn(N, generator=g).to(dev)
am, wm = ops.rowmax(A), ops.rowmax(W); planes = ops.presplit_w(W, wm)
f = lambda: ops.gemm_chain(A, W, b, relu=True, a_rowmax=am, w_rowmax=wm, w_planes=planes)
for _ in range(5): f()
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(50): f()
    torch.cuda.synchronize(); n += 50
print("  launches/s-derived us per launch: %.1f" % ((time.time() - t0) / n * 1e6), flush=True)
PY
  pid=$!
  sleep 4.5
  echo "== $tag"
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
  sleep 0.7
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
  wait $pid
}
run "product pp=1" LS_GEMM_W2_PP=1
run "product pp=0" LS_GEMM_W2_PP=0
for v in "$@"; do run "$v" LS_LIB_PATH=livingscenes_amd/lib/variants/$v/liblivingscenes_hip.so; done
