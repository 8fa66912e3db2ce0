#!/usr/bin/env python3
"""FlyingShape-style evaluation on synthetic scenes (counterpart of the reference's eval_flyingshape.py main, :216-232).
    python scripts/eval_flyingshape_synth.py --scenes 4 --objects 32 [--ckpt <dir with checkpoint/*latest.pt + files_backup/*.yaml>]
Without --ckpt the deterministic synthetic weights are used (the released checkpoint is not in the reference tree), so
accuracy numbers are those of an UNTRAINED network; the harness, metrics and throughput are what is exercised."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import harness, synth
from livingscenes_amd.lib_more.more_solver import More_Solver
from livingscenes_amd.model_utils import Shape_Prior

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=4); ap.add_argument("--objects", type=int, default=32)
ap.add_argument("--points", type=int, default=1024); ap.add_argument("--ckpt", default=None); ap.add_argument("--no-icp", action="store_true")
ap.add_argument("--data-root", default=None,
                help="FlyingShape directory tree (<root>/<.._n>/<scene>/*.npz, livingscenes_amd/datasets.py); if it does not exist the "
                     "synthetic scenes are written there first, then everything is read back from disk as the reference's loop does")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = {"shape_priors": {"n_input_point": 1024, "prior_name": "chair", "ckpt_dir": a.ckpt or ""}, "fps": {"n_init": 1, "random_start": False}}
if a.ckpt:
    solver = More_Solver(cfg)
else:
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
    solver = More_Solver(cfg, model=sp)
scenes = [synth.make_scene_pair(a.objects, a.points, seed=100 + i) for i in range(a.scenes)]
if a.data_root:
    from livingscenes_amd import datasets
    if not os.path.isdir(a.data_root):
        for i, sc in enumerate(scenes):
            datasets.write_scene(a.data_root, f"n_shape_{a.objects}", f"scene_{i:04d}",
                                 [{"pc": sc["ref"].numpy(), "transform": sc["ref_T"].numpy()},
                                  {"pc": sc["rescan"].numpy(), "transform": sc["rescan_T"].numpy()}])
    scenes = [datasets.scene_from_scans(data) for data in datasets.FlyingShape(a.data_root)]
t0 = time.perf_counter(); m = harness.eval_matching(scenes, solver); torch.cuda.synchronize(); t1 = time.perf_counter()
r = harness.eval_relocalization(scenes, solver, icp=not a.no_icp); torch.cuda.synchronize(); t2 = time.perf_counter()
r = {k: v for k, v in r.items() if not hasattr(v, "shape")}
print(json.dumps({"matching": m, "matching_s": t1 - t0, "relocalization": r, "relocalization_s": t2 - t1}, indent=1))
