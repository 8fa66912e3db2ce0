"""BASELINE.json configs[3] and configs[4] with the instances SHARDED over the GPUs of one node (livingscenes_amd/sharding.py,
lib_more.more_solver.solve_end2end_batch(sharded=True)), synthetic data, released widths:

    python scripts/configs_sharded.py [--gpus N] [--scenes 16] [--optim] [--dense-instances 256]           # N ranks, one per GPU (self-launching: livingscenes_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/configs_sharded.py ...

  configs[3]  3RScan-style end to end (eval_3rscan.py:337-463): 16 scenes x (reference + 2 rescans) x 8-24 instances of 1 024 .. 60 000 raw
              points; the flat (scene, instance) list is block-partitioned for ragged FPS + encode, the codes all-gathered over RCCL
              (4.1 KB per instance), the per-scene 32 x 32 matchers run replicated, the matched pairs are block-partitioned for Kabsch +
              ICP (or the 400-step optim refinement, --optim) and their (R | t) rows all-gathered (48 B per pair); every rank meshes
              the pairs of its block (MISE 32 -> 128 + marching cubes; grids and meshes stay local).
  configs[4]  dense SDF reconstruction (eval_3rscan.py:466-502): 128^3 query grid per instance, the instances block-partitioned, no
              exchange at all.
Timing: barrier + device sync on both sides, MAX over ranks; rank 0 prints one JSON line per config with the whole-job figures."""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from livingscenes_amd import launch

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=0, help="ranks (one per GPU); 0 = whatever the launcher started (1 when started plainly)")
ap.add_argument("--scenes", type=int, default=16)
ap.add_argument("--optim", action="store_true", help="registration.optim: true (the 400-step refinement of every matched pair)")
ap.add_argument("--optim-steps", type=int, default=400)
ap.add_argument("--no-mesh", action="store_true")
ap.add_argument("--dense-instances", type=int, default=256)
ap.add_argument("--skip-dense", action="store_true")
ap.add_argument("--skip-scenes", action="store_true")
args = ap.parse_args()
# --gpus N started plainly re-executes under torch.distributed.run with N ranks; under a launcher WORLD_SIZE must equal N
world, rank, local = launch.ensure_ranks(args.gpus or int(os.environ.get("WORLD_SIZE", "1")), __file__, sys.argv[1:])
import numpy as np, torch
import torch.distributed as dist
from livingscenes_amd import sharding, synth
from livingscenes_amd.lib_more.more_solver import More_Solver, solve_end2end_batch
from livingscenes_amd.model_utils import Shape_Prior
multi = world > 1
torch.cuda.set_device(local if torch.cuda.device_count() > local else 0)
dev = torch.device("cuda", torch.cuda.current_device())
if multi:
    dist.init_process_group(os.environ.get("LS_DIST_BACKEND", "nccl"), device_id=dev if os.environ.get("LS_DIST_BACKEND", "nccl") == "nccl" else None)
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
if multi:
    sharding.broadcast_weights(sp, src=0)        # 29.6 MB once (every rank built the same synthetic weights; this is the deployment's step)
cfg = {"shape_priors": {"n_input_point": 1024}, "fps": {"n_init": 1},
       "registration": {"step_size": {"so3": 0.05}, "n_steps": args.optim_steps, "early_stop_threshold": 10},
       "mesh_extractor": dict(threshold=0.5, resolution0=32, upsampling_steps=2, padding=0.1, points_batch_size=400000)}
solver = More_Solver(cfg, model=sp)


def timed(fn):
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return out, float(dt)


if not args.skip_scenes:
    rng = np.random.default_rng(0)
    scenes = []
    for s in range(args.scenes):
        n = int(rng.integers(8, 25))
        shapes = rng.integers(0, 10 ** 6, n)
        scenes.append([synth.make_raw_scan(shapes, 100 * s + k, device=dev)[0] for k in range(3)])   # reference + two rescans of the same objects
    pairs = [(sc[0], sc[k]) for sc in scenes for k in (1, 2)]
    code0 = sp.encode_fps(scenes[0][0]["pc"][:1], scenes[0][0]["pc_mask"][:1])
    canon = {k: v.clone() for k, v in code0.items()}
    canon["t"], canon["s"] = torch.zeros_like(canon["t"]), torch.ones_like(canon["s"])
    level = float(np.median(solver.mesh_extractor.eval_grid(canon, sp.decoder)))     # iso-level of the untrained field
    solver.mesh_extractor.threshold = 1.0 / (1.0 + np.exp(-level))
    solve_end2end_batch(solver, pairs[:2], mesh=not args.no_mesh, optim=False, sharded=multi)   # warm-up
    outs, dt = timed(lambda: solve_end2end_batch(solver, pairs, mesh=not args.no_mesh, optim=args.optim, sharded=multi))
    n_inst = sum(p[0]["pc"].shape[0] + p[1]["pc"].shape[0] for p in pairs)
    n_pairs = sum(r is not None for o in outs for r in o["registration"])
    n_mesh = torch.tensor([sum(m is not None for o in outs for m in o["mesh_lst"])], device=dev)
    if multi:
        dist.all_reduce(n_mesh)
    if rank == 0:
        print(json.dumps({"config": "configs[3] 3RScan-style end to end (synthetic)", "n_gpus": world, "scene_pairs": len(pairs), "instance_encodes": n_inst,
                          "registrations": n_pairs, "optim": bool(args.optim), "meshes": int(n_mesh), "seconds": round(dt, 3),
                          "matched_objects_per_s": round(n_pairs / dt, 2)}))

if not args.skip_dense:
    G = 128
    lin = (torch.arange(G, device=dev, dtype=torch.float32) + 0.5) / G - 0.5
    grid = (1.1 * torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3))
    n = args.dense_instances
    x = synth.make_instances(min(n, 64), 1024, seed=3)
    x = (x if isinstance(x, torch.Tensor) else x[0]).to(dev)
    base = sp.encode(x)
    codes = {k: v.repeat((n + v.shape[0] - 1) // v.shape[0], *([1] * (v.dim() - 1)))[:n].contiguous() for k, v in base.items()}
    sp.decoder(grid[:, :65536].expand(2, -1, -1).contiguous(), None, {k: v[:2] for k, v in codes.items()}, return_sdf=True)   # warm-up

    def dense():
        lo, hi = sharding.shard_range(n, rank, world)
        for b0 in range(lo, hi, 8):                                          # 8 instances x 128^3 per decoder call (the workspace bound)
            b1 = min(hi, b0 + 8)
            sp.decoder(grid.expand(b1 - b0, -1, -1).contiguous(), None, {k: v[b0:b1] for k, v in codes.items()}, return_sdf=True)
        return hi - lo
    _, dt = timed(dense)
    if rank == 0:
        nq = n * G ** 3
        print(json.dumps({"config": "configs[4] dense SDF 128^3", "n_gpus": world, "instances": n, "seconds": round(dt, 3),
                          "Mqueries_per_s": round(nq / dt / 1e6, 2)}))
if multi:
    dist.destroy_process_group()
