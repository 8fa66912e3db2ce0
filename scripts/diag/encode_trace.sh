#!/bin/bash
# Per-kernel durations of plain sequential encodes (one step in flight, ONE stream: LS_FPS_SIDE=0) under rocprofv3, fused vs unfused
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/enc_trace_$f; rm -rf $out; mkdir -p $out
  LS_EDGE_FUSE_Q=$f LS_FPS_SIDE=0 rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python - > $out/log.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from livingscenes_amd import synth
from livingscenes_amd.model_utils import Shape_Prior
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
x = synth.make_instances(64, 1024, seed=1000)
x = (x if isinstance(x, torch.Tensor) else x[0]).to(dev)
for _ in range(20):
    sp.encode(x)
    torch.cuda.synchronize()
PY
  python - $out $f <<'PY'
import csv, glob, sys
out, f = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0])))
sel = [r for r in rows if "edge_attn" in r["Name"] or "gemm_" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"fuse {f}: total kernel time per encode {tot/20/1e6:.3f} ms")
for r in sorted(sel, key=lambda r: -float(r["TotalDurationNs"])): print(f'   {r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
done
