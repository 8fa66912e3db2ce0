"""BASELINE.json configs[3] and configs[4] on ONE MI355X with synthetic data (the 8-GPU runs shard instances / scenes across ranks
with no data-path collective, livingscenes_amd/sharding.py):

  configs[3]  3RScan-style end to end: 16 scenes x (1 reference + 2 rescans) x 8-24 instances of 1 024 .. 60 000 raw points ->
              ragged FPS + encode, sequential matching, Kabsch + ICP registration of the matched pairs, SDF reconstruction
              (MISE 32 -> 128 + marching cubes) of every transformed code   (More_Solver._solve_end2end, more_solver.py:246-299;
              + the Adam refinement of eval_3rscan's optim=True on every matched pair, 64 pairs in lock-step)
  configs[4]  dense SDF reconstruction: 128^3 query grid per instance, 256 instances, decoder GEMM path
"""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from livingscenes_amd import synth
from livingscenes_amd.lib_more.more_solver import More_Solver
from livingscenes_amd.model_utils import Shape_Prior

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=16)
ap.add_argument("--dense-instances", type=int, default=256)
ap.add_argument("--skip-dense", action="store_true")
ap.add_argument("--optim-pairs", type=int, default=-1, help="matched pairs put through the optim=True refinement (-1 = all, 0 = skip)")
ap.add_argument("--optim-chunk", type=int, default=128, help="pairs advanced in lock-step per call")
args = ap.parse_args()
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
cfg = {"shape_priors": {"n_input_point": 1024}, "fps": {"n_init": 1},
       "registration": {"step_size": {"so3": 0.05}, "n_steps": 400, "early_stop_threshold": 10},
       "mesh_extractor": dict(threshold=0.5, resolution0=32, upsampling_steps=2, padding=0.1, points_batch_size=400000)}
solver = More_Solver(cfg, model=sp)
rng = np.random.default_rng(0)


def raw_scan(shapes, seed):
    """Instances as raw clouds of 1 024 .. 60 000 points (re-sampled canonical shapes under a random rigid motion), padded."""
    r = np.random.default_rng(seed)
    clouds = []
    for s in shapes:
        P = int(np.exp(r.uniform(np.log(1024), np.log(60000))))
        c = synth.canonical_shape(P, int(s))
        c = torch.as_tensor(c, dtype=torch.float32)
        Rm = torch.as_tensor(synth._rand_rot(r), dtype=torch.float32)
        clouds.append(c @ Rm.T + torch.as_tensor(r.uniform(-2, 2, 3), dtype=torch.float32))
    mx = max(c.shape[0] for c in clouds)
    pc = torch.zeros(len(clouds), 3, mx)
    mask = torch.zeros(len(clouds), 1, mx, dtype=torch.bool)
    for i, c in enumerate(clouds):
        pc[i, :, :c.shape[0]] = c.T
        mask[i, :, :c.shape[0]] = True
    return {"pc": pc.to(dev), "pc_mask": mask.to(dev)}, sum(c.shape[0] for c in clouds)


# ---------------------------------------------------------------------------------------------------------- configs[3]
scenes = []
for s in range(args.scenes):
    n = int(rng.integers(8, 25))
    shapes = rng.integers(0, 10 ** 6, n)
    scenes.append([raw_scan(shapes, 100 * s + k) for k in range(3)])   # reference + two rescans of the same objects
n_inst = sum(sc[0][0]["pc"].shape[0] for sc in scenes) * 3
n_pts = sum(p for sc in scenes for _, p in sc)
solver.mesh_extractor.threshold = 0.5
# iso-level: the synthetic (untrained) field has no zero level set; cut it at the median logit of one canonical code
code0 = sp.encode_fps(scenes[0][0][0]["pc"][:1], scenes[0][0][0]["pc_mask"][:1])
canon = {k: v.clone() for k, v in code0.items()}
canon["t"], canon["s"] = torch.zeros_like(canon["t"]), torch.ones_like(canon["s"])
level = float(np.median(solver.mesh_extractor.eval_grid(canon, sp.decoder)))
solver.mesh_extractor.threshold = 1.0 / (1.0 + np.exp(-level))
solver._solve_end2end(scenes[0][0][0], scenes[0][1][0], mesh=True)   # warm-up
torch.cuda.synchronize()
t = {"encode+match+register": 0.0, "mesh": 0.0}
n_pairs = n_mesh = 0
for sc in scenes:
    ref = sc[0][0]
    for k in (1, 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = solver._solve_end2end(ref, sc[k][0], mesh=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for c in out["codes"]:
            if c is not None:
                solver._mesh_from_latent(c); n_mesh += 1
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t["encode+match+register"] += t1 - t0
        t["mesh"] += t2 - t1
        n_pairs += sum(r is not None for r in out["registration"])
tot = sum(t.values())
enc_inst = sum(2 * sc[0][0]["pc"].shape[0] for sc in scenes) * 2      # both scans of a pair are encoded, two pairs per scene
print(f"configs[3] synthetic, 1 GPU: {args.scenes} scenes x 3 scans, {n_inst} instances, {n_pts/1e6:.1f} M raw points; "
      f"{enc_inst} instance encodes (ragged FPS + encoder), {n_pairs} registrations (Kabsch + ICP), {n_mesh} meshes (129^3 via MISE + MC)")
print(f"  encode+match+register {t['encode+match+register']:.2f} s ({enc_inst / t['encode+match+register']:.0f} instance-encodes/s), "
      f"SDF reconstruction {t['mesh']:.2f} s ({t['mesh'] / max(n_mesh, 1) * 1e3:.1f} ms per mesh), total {tot:.2f} s = "
      f"{n_pairs / tot:.1f} matched objects/s end to end")
# the same work with every scan / matched pair of all scenes batched (lib_more.more_solver.solve_end2end_batch)
from livingscenes_amd.lib_more.more_solver import solve_end2end_batch
all_pairs = [(sc[0][0], sc[k][0]) for sc in scenes for k in (1, 2)]
solve_end2end_batch(solver, all_pairs[:2])
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = solve_end2end_batch(solver, all_pairs)
torch.cuda.synchronize(); tb = time.perf_counter() - t0
nb = sum(r is not None for o in outs for r in o["registration"])
print(f"  batched over all {len(all_pairs)} scene pairs: encode+match+register {tb:.2f} s ({enc_inst / tb:.0f} instance-encodes/s, {nb} registrations)")
solve_end2end_batch(solver, all_pairs[:1], mesh=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
outs = solve_end2end_batch(solver, all_pairs, mesh=True)
torch.cuda.synchronize(); tm = time.perf_counter() - t0
nm = sum(m is not None for o in outs for m in o["mesh_lst"])
print(f"  batched incl. SDF reconstruction (16 MISE octrees in lock-step, ragged decoder calls): {tm:.2f} s total, "
      f"{(tm - tb) / max(nm, 1) * 1e3:.1f} ms per mesh, {nm / tm:.1f} matched-and-meshed objects/s")
# eval_3rscan's optim=True refinement (400 Adam steps per matched pair, more_solver.py:118-189; eval_3rscan.py:381) of EVERY matched pair,
# 64 pairs in lock-step per call (csrc/optim.hip); round 1 ran it pair by pair at 1.1 s per pair
reg1 = [o["ref_pc_lst"][i] for o in outs for i, j in enumerate(o["matches"].tolist()) if j >= 0]
reg2 = [o["rescan_pc_lst"][j] for o in outs for i, j in enumerate(o["matches"].tolist()) if j >= 0]
if args.optim_pairs >= 0:
    reg1, reg2 = reg1[:args.optim_pairs], reg2[:args.optim_pairs]
if reg1:
    solver._solve_pairwise_registration_optim_batch(reg1[:2], reg2[:2])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for c0 in range(0, len(reg1), args.optim_chunk):
        solver._solve_pairwise_registration_optim_batch(reg1[c0:c0 + args.optim_chunk], reg2[c0:c0 + args.optim_chunk])
    torch.cuda.synchronize(); to = time.perf_counter() - t0
    print(f"  optim=True refinement of {len(reg1)} matched pairs, {args.optim_chunk} in lock-step per call (400 Adam steps each: decoder forward + "
          f"backward, Sinkhorn, SE(3) Adam on the device): {to:.1f} s = {to / len(reg1) * 1e3:.1f} ms per pair")

# ---------------------------------------------------------------------------------------------------------- configs[4]
if not args.skip_dense:
    G = 128
    lin = (torch.arange(G, device=dev, dtype=torch.float32) + 0.5) / G - 0.5
    grid = (1.1 * torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3))
    x = synth.make_instances(8, 1024, seed=3)
    x = (x if isinstance(x, torch.Tensor) else x[0]).to(dev)
    codes = sp.encode(x)
    q = grid.expand(8, -1, -1).contiguous()
    sp.decoder(q[:, :65536], None, codes, return_sdf=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    done = 0
    while done < args.dense_instances:
        sp.decoder(q, None, codes, return_sdf=True)
        done += 8
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nq = done * G ** 3
    print(f"configs[4], 1 GPU: {done} instances x 128^3 = {nq/1e6:.0f} M queries in {dt:.1f} s = {nq/dt/1e6:.1f} M queries/s "
          f"({nq/dt*6.7e-6:.0f} TFLOP/s fp32-equivalent at 6.7 MFLOP per query = {nq/dt*20.1e-6:.0f} TFLOP/s of f16 MFMA as executed, 3 per fp32 product = "
          f"{nq/dt*20.1e-6/2500*100:.0f} % of the 2.5 PFLOP/s matrix peak); "
          f"an 8-GPU node shards the instances: {dt/8:.1f} s")
