#!/bin/bash
# SDF reconstruction (MISE + decoder + marching cubes) of 40 instances: wall time vs summed kernel time (is the host the bound?)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/mesh_prof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python - > $out/log.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from livingscenes_amd import synth
from livingscenes_amd.model_utils import Shape_Prior
from livingscenes_amd.lib_more.more_solver import More_Solver
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
cfg = {"shape_priors": {"n_input_point": 1024}, "fps": {"n_init": 1}, "registration": {"step_size": {"so3": 0.05}, "n_steps": 400, "early_stop_threshold": 10},
       "mesh_extractor": dict(threshold=0.5, resolution0=32, upsampling_steps=2, padding=0.1, points_batch_size=400000)}
solver = More_Solver(cfg, model=sp)
x = synth.make_instances(40, 1024, seed=3)
x = (x if isinstance(x, torch.Tensor) else x[0]).to(dev)
codes = sp.encode(x)
one = lambda i: {k: v[i:i + 1] for k, v in codes.items()}
canon = one(0)
level = float(np.median(solver.mesh_extractor.eval_grid(canon, sp.decoder)))
solver.mesh_extractor.threshold = 1.0 / (1.0 + np.exp(-level))
solver._mesh_from_latent(one(0))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(40): solver._mesh_from_latent(one(i))
torch.cuda.synchronize(); print("WALL_MS_PER_MESH", (time.perf_counter() - t0) / 40 * 1e3)
PY
grep WALL $out/log.txt
python - $out <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"summed kernel time per mesh (41 meshes + encode): {tot/41/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]: print(f'   {r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.1f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
