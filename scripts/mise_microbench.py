"""MISE-driven SDF grid (BASELINE configs[3]-style decode, SURVEY 8 f-2 first half): Generator3D.eval_grid per instance at the
released extraction settings (resolution0 32, two up-sampling steps -> 129^3 lattice, threshold 0.5, padding 0.1) vs the dense
129^3 evaluation of the same lattice."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from livingscenes_amd import synth
from livingscenes_amd.mesh_extractor2 import Generator3D
from livingscenes_amd.model_utils import Shape_Prior
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
x = synth.make_instances(8, 1024, seed=0)
x = (x if isinstance(x, torch.Tensor) else x[0]).to(dev)
emb = sp.encode(x)
# the synthetic (untrained) weights give a field without a zero level set inside the box: put the iso-level at the median
# logit of instance 0 on a coarse lattice so that MISE has a surface to refine
G = 129
lin = torch.arange(G, device=dev, dtype=torch.float32) / 128.0
pts = (1.1 * (torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3) - 0.5))[None]
code = {k: v[:1] for k, v in emb.items()}
dense = -sp.decoder(pts, None, code, return_sdf=True)[0]
level = float(dense.median())
thr = 1.0 / (1.0 + np.exp(-level))
gen = Generator3D(threshold=thr, resolution0=32, upsampling_steps=2, padding=0.1, points_batch_size=400000)
for b in range(2):
    gen.eval_grid(code, sp.decoder)
torch.cuda.synchronize()
t0 = time.perf_counter(); nq = []
for b in range(8):
    st = {}
    g = gen.eval_grid(code, sp.decoder, stats_dict=st)
    nq.append(st["mise rounds"])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
tot = np.mean([sum(r) for r in nq])
print(f"MISE grid 129^3 (iso-level = median logit {level:.4f}): {dt*1e3:.1f} ms / instance, rounds {nq[0]}, queries {tot:.0f} = "
      f"{100*tot/129**3:.1f} % of dense -> {tot/dt/1e6:.1f} M queries/s")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): d = sp.decoder(pts, None, code, return_sdf=True)
torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 3
print(f"dense 129^3: {dd*1e3:.1f} ms / instance ({G**3/dd/1e6:.1f} M queries/s)")
evald = torch.from_numpy(g).to(dev).reshape(-1).float()
same = ((-d[0]) == evald)
print(f"MISE grid == dense grid on {float(same.float().mean())*100:.1f} % of the lattice (the rest is completed, not evaluated); "
      f"sign agreement w.r.t. the iso-level: {float((((-d[0]) > level) == (evald > level)).float().mean())*100:.2f} %")
# marching cubes on the padded 131^3 grid (Generator3D.extract_mesh)
gen.extract_mesh(g, None, code); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): mesh = gen.extract_mesh(g, None, code)
torch.cuda.synchronize(); dm = (time.perf_counter() - t0) / 5
from livingscenes_amd.mesh_extractor2 import marching_cubes
vol = torch.nn.functional.pad(torch.from_numpy(g).to(dev), (1, 1, 1, 1, 1, 1), value=-1e6)
marching_cubes(vol, level); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): v, f = marching_cubes(vol, level)
torch.cuda.synchronize(); dk = (time.perf_counter() - t0) / 5
print(f"marching cubes 131^3: {dk*1e3:.2f} ms on the device (two passes incl. the count read-back), extract_mesh incl. host copies "
      f"{dm*1e3:.1f} ms; mesh {len(mesh.vertices)} vertices / {len(mesh.faces)} faces")
