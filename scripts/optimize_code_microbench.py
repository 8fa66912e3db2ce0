"""More_Solver._optimize_code (code half): 200 Adam steps on one instance code against the SDF at 1024 observed
points, released decoder (9 x 768); decoder forward + backward in the HIP library, Adam in torch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import synth
from livingscenes_amd.lib_more.more_solver import More_Solver
from livingscenes_amd.model_utils import Shape_Prior
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
x = synth.make_instances(1, 1024, seed=0)
x = (x if isinstance(x, torch.Tensor) else x[0]).to(dev)
code = sp.encode(x)
solver = More_Solver({"shape_priors": {"n_input_point": 1024}}, model=sp)
mask = torch.ones(1, 1024, dtype=torch.bool, device=dev)
with torch.no_grad():
    l0 = float(sp.decoder(x.transpose(1, 2), None, code, return_sdf=True).pow(2).mean())
solver._optimize_code({k: v.detach().clone() for k, v in code.items()}, x[0], mask, n_steps=5)
torch.cuda.synchronize(); t0 = time.perf_counter()
best = solver._optimize_code({k: v.detach().clone() for k, v in code.items()}, x[0], mask, n_steps=200)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
with torch.no_grad():
    l1 = float(sp.decoder(x.transpose(1, 2), None, best, return_sdf=True).pow(2).mean())
fl = 3 * 2 * 1024 * (768 * 768 * 6 + 768 * 256 * 2 + 768 * 8) * 200 / 1e12
print(f"_optimize_code: 200 steps in {dt*1e3:.1f} ms = {dt*5:.3f} ms/step (fwd + bwd + Adam), ~{fl/dt:.1f} TFLOP/s on the 768-wide layers; "
      f"MSE(sdf) {l0:.3e} -> {l1:.3e}")
