"""More_Solver._solve_pairwise_registration(optim=True) (registration half; manifold Adam and Sinkhorn follow this
build's definitions, parity unpinned): released settings (400 steps, lr 0.05, 1024 points, released decoder) on synthetic pairs,
P pairs in lock-step (csrc/optim.hip).  usage: python scripts/optim_registration_microbench.py [P ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import synth
from livingscenes_amd.lib_math.torch_se3 import concatenate, inverse
from livingscenes_amd.lib_more.more_solver import More_Solver
from livingscenes_amd.lib_more.pose_estimation import rotation_error, translation_error
from livingscenes_amd.model_utils import Shape_Prior
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
cfg = {"shape_priors": {"n_input_point": 1024}, "fps": {"n_init": 1},
       "registration": {"step_size": {"so3": 0.05}, "n_steps": 400, "early_stop_threshold": 10}}
solver = More_Solver(cfg, model=sp)
if os.environ.get("PIECES"):
    cfg["registration"]["decoder_bf16_pieces"] = int(os.environ["PIECES"])
for P in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    sc = synth.make_scene_pair(P, 1024, seed=5, noise=0.005)
    gt = concatenate(sc["rescan_T"][:, :3], inverse(sc["ref_T"][:, :3])).to(dev)
    p1, p2 = [sc["ref"][i].to(dev) for i in range(P)], [sc["rescan"][i].to(dev) for i in range(P)]
    R0, t0 = solver._solve_pairwise_registration_batch(p1, p2)
    torch.cuda.synchronize(); t_0 = time.perf_counter()
    R1, t1, info = solver._solve_pairwise_registration_optim_batch(p1, p2, return_info=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t_0
    e0, e1 = rotation_error(R0, gt[:, :, :3]).reshape(-1), rotation_error(R1, gt[:, :, :3]).reshape(-1)
    print(f"P = {P}: {dt:.2f} s for {info['steps']} steps = {dt / info['steps'] * 1e3:.2f} ms/step = {dt / P * 1e3:.1f} ms per pair; "
          f"median RRE Kabsch+ICP {float(e0.median()):.3f} deg -> optim+ICP {float(e1.median()):.3f} deg; "
          f"median RTE {float(translation_error(t1, gt[:, :, 3:4]).median()) * 1e3:.2f} mm; pairs stopped early {int((info['active'] == 0).sum())}")
