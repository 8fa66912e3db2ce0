"""SDF decoder throughput (BASELINE.json configs[4] style: dense query grids), queries/s and effective TFLOP/s."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from livingscenes_amd import synth
from livingscenes_amd.model_utils import Shape_Prior
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8); ap.add_argument("--res", type=int, default=64); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
emb = sp.encode(synth.make_instances(a.B, 1024, seed=0).to(dev))
M = a.res ** 3
lin = torch.linspace(-0.55, 0.55, a.res, device=dev)
grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, M, 3)
q = grid.expand(a.B, -1, -1) * emb["s"][:, None, None] + emb["t"]
sp.decoder(q[:, :4096].contiguous(), None, emb, return_sdf=True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    sdf = sp.decoder(q.contiguous(), None, emb, return_sdf=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
nq = a.B * M
print(f"sdf decode B={a.B} grid={a.res}^3: {dt*1e3:.2f} ms  {nq/dt/1e6:.2f} Mqueries/s  {nq*6.7e6/dt/1e12:.1f} TFLOP/s (6.7 MFLOP/query as executed; reference formulation 8.26)")
