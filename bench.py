#!/usr/bin/env python3
"""bench.py -- object-instances/sec for the LivingScenes per-instance hot path (encode + match + register) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One STEP (per rank) = one pass of the hot path over one batch of synthetic input already resident in HBM:
  BASELINE.json configs[1] + [2]: 64 instances x 1024 points (a 32-object reference scene and its 32-object rescan)
    -> Shape_Prior.encode (prologue, FPS, 7x [k-NN, VN edge-conv], heads)            (64 objects)
    -> sequential_matcher on the 32x32 invariant-code scores                          (1 scene pair)
    -> kabsch_transformation_estimation on the 32 matched pairs (z_so3 + t)           (32 poses)
Weights: deterministic synthetic weights of the released architecture (the checkpoint is absent from the reference
tree); data: synthetic chair-like clouds (livingscenes_amd.synth).  fp32 throughout, as the reference.
Multi-GPU: instances shard embarrassingly (weak scaling: every rank runs its own 64-instance batch); RCCL is used to
broadcast the weights once and to gather a result checksum -- there is no data-path collective.

Prints ONE JSON line (rank 0): metric/value/... + "roofline" (dominant kernel, hipEvent-timed per launch in a separate
profiled pass of the same K steps) + "cpu_baseline" (the CPU oracle = the reference's PyTorch-CPU op sequence, timed on
this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

# Each in-flight step drives one caller stream plus two library side streams; ROCm maps HIP streams round-robin onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and kernels that share a hardware queue serialise.  Measured on MI355X
# (DESIGN.md 6): 4 queues / 3 steps in flight 25.9k obj/s, 16 queues / 8 steps 28.8k, 32 queues worse.  Must be set before the
# HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # fp32 vector == fp32 MFMA peak


def layer_plan(cfg, N):
    """(Ns, Nd, Cin, Co, attn, glob) per encoder layer."""
    out, cur = [], N
    for i in range(cfg["num_layers"]):
        ns = cur
        if i in cfg["down_sample_layers"]:
            cur //= cfg["down_sample_factor"][cfg["down_sample_layers"].index(i)]
        out.append(dict(Ns=ns, Nd=cur, Cin=1 if i == 0 else cfg["feat_dim"][i - 1], Co=cfg["feat_dim"][i],
                        attn=i >= cfg["atten_start_layer"], glob=i >= cfg["res_global_start_layer"]))
    return out


def committed_pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_latest.json, produced by
    scripts/pmc_summary.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench command; counters cannot be
    read from inside the process).  None when no committed measurement names this kernel."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            pmc = json.load(f)
        e = pmc[kernel]
        return e["hbm_read_bytes"] + e["hbm_write_bytes"], "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
    except (OSError, KeyError, ValueError):
        return None, None


def algorithmic_cost(kind, layer, cfg, B, N):
    """(bytes, flops) one launch of kernel `kind` at `layer` must move / execute."""
    pl = layer_plan(cfg, N)
    L = pl[min(layer, len(pl) - 1)]
    Ns, Nd, Cin, Co = L["Ns"], L["Nd"], L["Cin"], L["Co"]
    f4 = 4
    if kind == "knn":
        D = 3 * Cin
        return B * ((Nd + Ns) * D * f4 + Nd * 16 * 4), 3.0 * B * Nd * Ns * D
    if kind == "gemm_edge":
        nc = (10 if L["attn"] else 4) * Co
        pc = (4 if L["attn"] else 2) * Co
        if Nd != Ns:  # down-sampled layer: neighbour-side columns on Ns rows, destination-side columns on Nd rows
            rows_cols = Ns * pc + Nd * (nc - pc)
            return B * 3 * (Ns * Cin + Nd * Cin + rows_cols) * f4 + nc * Cin * f4, 2.0 * B * 3 * Cin * rows_cols
        return B * Ns * 3 * (Cin + nc) * f4 + nc * Cin * f4, 2.0 * B * Ns * 3 * Cin * nc
    if kind == "edge_attn":
        gather = B * Nd * 16 * 4 * Co * 3 * f4          # P_lin/P_dir of the K and V branches at 16 neighbours
        own = B * Nd * 6 * Co * 3 * f4                  # Q-side columns of the destination row
        return gather + own + B * Nd * 16 * 4 + B * Nd * 3 * Co * f4, 2.0 * B * Nd * 16 * Co * 2 * 30
    if kind == "edge_pool":
        return B * Nd * 16 * 2 * Co * 3 * f4 + B * Nd * 2 * Co * 3 * f4 + B * Nd * 16 * 4 + B * Nd * 3 * Co * f4, 2.0 * B * Nd * 16 * Co * 30
    if kind == "edge_l0":
        return B * Nd * 16 * (12 + 4) + B * Nd * 12 + B * Nd * 3 * Co * f4, 2.0 * B * Nd * 16 * Co * 40
    if kind == "gemm_glob":
        return B * Nd * 3 * 3 * Co * f4 + 4 * Co * Co * f4, 2.0 * B * Nd * 3 * Co * 2 * Co
    if kind == "vn_act":
        return B * Nd * 3 * 3 * Co * f4, 30.0 * B * Nd * Co
    if kind == "mean":
        return B * Nd * 3 * Co * f4, 1.0 * B * Nd * 3 * Co
    if kind == "fps":
        n = [N] + [p["Nd"] for p in pl if p["Nd"] != p["Ns"]]
        return B * n[min(layer, len(n) - 1)] * 12, 0.0
    if kind == "prologue":
        return B * N * 24, 8.0 * B * N * N / 2
    if kind == "gemm_tail":
        return B * pl[-1]["Nd"] * 3 * (pl[-1]["Co"] + cfg["c_dim"]) * f4, 2.0 * B * pl[-1]["Nd"] * 3 * pl[-1]["Co"] * cfg["c_dim"]
    if kind == "tail":
        c = cfg["c_dim"]
        return B * pl[-1]["Nd"] * 3 * c * f4 + 2 * c * c * f4, 2.0 * B * 3 * c * c * 2
    return 0, 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=480,
                    help="timed steps (a step is ~2.5 ms: a 24-step region was short enough for one host hiccup to cost 25 %%)")
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--batch", type=int, default=64, help="instances per step per GPU (two scenes of batch/2 objects)")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--cpu-instances", type=int, default=8, help="bounded sample for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads for the CPU baseline: the reference's op sequence peaks at ~16 threads on the GPU box's "
                         "2x64-core host (0.65 inst/s at 16 vs 0.25 at 128 vs 0.07 at 256; tests/tools/cpu_threads_probe.py)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-fma-variant", action="store_true", help="skip the secondary fused-multiply-add k-NN timing")
    ap.add_argument("--inflight", type=int, default=8,
                    help="independent steps kept in flight on separate HIP streams (each with its own model handle and "
                         "workspace): one step's low-occupancy kernels (FPS, heads, matcher) overlap another's big ones")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the hot path)")
    # LS_BENCH_BACKEND=gloo: dry run of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices; the
    # collectives go through the host) -- a logic check only, never a measurement
    backend = os.environ.get("LS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from livingscenes_amd import sharding as parallel, synth
    from livingscenes_amd.lib_more.matcher_new import sequential_matcher
    from livingscenes_amd.lib_more.pose_estimation import kabsch_transformation_estimation
    from livingscenes_amd.model_utils import Shape_Prior

    if args.inflight > 1:
        os.environ.setdefault("LS_GEMM_OVERLAP", "0")
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    if rank == 0:
        ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    else:  # other ranks start from zeros and receive rank 0's weights over RCCL
        ew = {k: torch.zeros(s) for k, s in synth.encoder_param_shapes(ecfg).items()}
        dw = {k: torch.zeros_like(v) for k, v in synth.make_decoder_weights(dcfg, 0).items()}
    sp = Shape_Prior.from_state(ecfg, dcfg, ew, dw, device=dev)
    if world > 1:
        parallel.broadcast_weights(sp, src=0)
    nfl = max(1, args.inflight)
    if nfl > 1:
        # A/B on MI355X (scripts in DESIGN.md 6): intra-step GEMM||k-NN stream overlap is +5 % for a single in-flight step but
        # -3.5 % once two whole steps already overlap; the library default stays on, the bench turns it off.
        os.environ.setdefault("LS_GEMM_OVERLAP", "0")
    # one model handle (packed weights + side stream + workspace) per in-flight step; weights are shared tensors
    sps = [sp] + [Shape_Prior.from_state(ecfg, dcfg, sp.encoder.state_dict(), sp.decoder.F.state_dict(), device=dev) for _ in range(nfl - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]

    B, N = args.batch, args.points
    n_obj = B // 2
    scene = synth.make_scene_pair(n_obj, N, seed=1000 + rank)
    x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)  # [B,3,N] resident in HBM

    def step(sp=sp):
        emb = sp.encode(x)
        m = sequential_matcher(emb["z_inv"][:n_obj], emb["z_inv"][n_obj:])
        j = m["matches0"].clamp(min=0)
        p1 = emb["z_so3"][:n_obj] + emb["t"][:n_obj]
        p2 = (emb["z_so3"][n_obj:] + emb["t"][n_obj:]).index_select(0, j)
        R, t, _, _ = kabsch_transformation_estimation(p1, p2)
        return emb, m, R, t

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()

    def run(n):
        out = None
        for i in range(n):
            with torch.cuda.stream(streams[i % nfl]):
                out = step(sps[i % nfl])
        return out

    with torch.no_grad():
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        out = run(max(args.warmup, nfl))
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run(args.steps)
        dt_host = time.perf_counter() - t0     # host time to enqueue the K steps (no device sync inside a step)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())

    # secondary figure (never `value`): the same K steps with LS_FLAG_CONTRACT_FMA, i.e. dist = fmaf(diff, diff, dist) as nvcc
    # compiles pytorch3d's knn.cu -- the rounding the reference's CUDA deployment runs with
    dt_fma = None
    if not args.no_fma_variant:
        for s_ in sps:
            s_.knn_flags = 1
        with torch.no_grad():
            run(nfl)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            barrier()
            dt_fma = time.perf_counter() - t0
        for s_ in sps:
            s_.knn_flags = 0
        dt_f = torch.tensor([dt_fma], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt_f, op=dist.ReduceOp.MAX)
        dt_fma = float(dt_f.item())

    emb, m, R, t = out
    # sanity of the measured work (outside the timed region): matches are the identity permutation, poses are rotations
    n_correct = int((m["matches0"].cpu() == torch.arange(n_obj)).sum())
    det_ok = bool((torch.det(R.cpu()) > 0.99).all())
    if world > 1:  # RCCL all-gather of the per-rank codes (4.1 KB each: latency-bound; outside the timed region)
        allc = parallel.all_gather_codes(emb)
        assert allc["z_inv"].shape[0] == B * world

    roof = None
    if rank == 0 and not args.no_profile:
        hip = sp.hip_model()
        prof_steps = min(args.steps, 24)      # per-launch hipEvent pairs: a bounded, serial pass on one stream
        hip.profile_begin()
        with torch.no_grad():
            for _ in range(prof_steps):
                step()
        prof = hip.profile_end()
        tot = sum(p["total_ms"] for p in prof)
        by = sorted(prof, key=lambda p: -p["total_ms"])
        # dominant kernel = the largest launch of the kernel family with the largest share of device time
        fam = {}
        for q in prof:
            fam[q["kind"]] = fam.get(q["kind"], 0.0) + q["total_ms"]
        dom_kind = max(fam, key=fam.get)
        dom = max((q for q in prof if q["kind"] == dom_kind), key=lambda q: q["total_ms"])
        abytes, aflops = algorithmic_cost(dom["kind"], dom["layer"], ecfg, B, N)
        avg_s = dom["total_ms"] / dom["launches"] * 1e-3
        t_hbm, t_fl = abytes / (HBM_PEAK_GBS * 1e9), aflops / (FP32_PEAK_TFLOPS * 1e12)
        if t_fl >= t_hbm:
            pipe = {"gemm_edge": "mfma-f32", "gemm_glob": "mfma-f32",
                    "knn": "valu-f32 exact distances (+ bf16-mfma safe filter on the seeded layers)"}.get(dom["kind"], "valu-f32")
            roof = dict(bound="mfma", pipe=pipe, achieved=aflops / avg_s / 1e12, peak=FP32_PEAK_TFLOPS, unit="TFLOP/s")
            if dom["kind"] == "knn":
                roof["note"] = ("one k-NN graph build = the launch sequence of that layer (hints / centre / bf16 image / seed / sweep / "
                                "finish where seeded); achieved = algorithmic 3*Nd*Ns*3C flops of the direct-difference form / its duration")
        else:
            roof = dict(bound="hbm", achieved=abytes / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["kernel"] = f"{dom['kind']}[layer {dom['layer']}]"
        roof["traffic"], roof["traffic_source"] = committed_pmc_traffic(roof["kernel"])
        roof["avg_launch_us"] = avg_s * 1e6
        roof["algorithmic_bytes_per_launch"] = abytes
        roof["algorithmic_flops_per_launch"] = aflops
        roof["timing"] = f"hipEvent pair per launch on the launching stream, separate profiled pass of {prof_steps} steps"
        roof["share_of_device_time"] = dom["total_ms"] / max(tot, 1e-9)
        kinds = {}
        for p in prof:
            kinds[p["kind"]] = kinds.get(p["kind"], 0.0) + p["total_ms"]
        roof["breakdown_ms_per_step"] = {k: round(v / prof_steps, 4) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])}
        roof["per_layer_ms_per_step"] = {f"{p['kind']}{p['layer']}": round(p["total_ms"] / prof_steps, 4) for p in by[:40]}
        # the other two kernel families north_star names, on their largest launch: the VN edge-conv gather kernel
        # (HBM/L2-gather-bound) and the fp32-MFMA VN-Linear GEMM
        extra = []
        for kind in ("edge_attn", "gemm_edge"):
            cands = [p for p in prof if p["kind"] == kind]
            if not cands:
                continue
            e = max(cands, key=lambda p: p["total_ms"])
            eb, ef = algorithmic_cost(kind, e["layer"], ecfg, B, N)
            es = e["total_ms"] / e["launches"] * 1e-3
            hb = eb / (HBM_PEAK_GBS * 1e9) >= ef / (FP32_PEAK_TFLOPS * 1e12)
            extra.append(dict(kernel=f"{kind}[layer {e['layer']}]", bound="hbm" if hb else "mfma", avg_launch_us=es * 1e6,
                              achieved=(eb / es / 1e9) if hb else (ef / es / 1e12), peak=HBM_PEAK_GBS if hb else FP32_PEAK_TFLOPS,
                              unit="GB/s" if hb else "TFLOP/s", frac=((eb / es / 1e9) / HBM_PEAK_GBS) if hb else ((ef / es / 1e12) / FP32_PEAK_TFLOPS),
                              note="algorithmic gather bytes; > HBM peak means the gather is served by L2 / Infinity Cache" if kind == "edge_attn" else
                                   "table GEMM: fp32 MFMA, output write included in the algorithmic bytes"))
        roof["other_kernels"] = extra

    cpu = None
    if rank == 0 and world == 1 and args.cpu_instances > 0:
        from oracle import more, net  # the checker, timed as the CPU baseline ("port" of the reference's op sequence)
        nb = args.cpu_instances
        torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
        xc = x[:nb].cpu()
        ewc, _ = synth.make_encoder_weights(ecfg, 0), None
        t1 = time.perf_counter()
        with torch.no_grad():
            embc = net.shape_prior_encode(ewc, ecfg, xc)
            h = nb // 2
            mm = more.sequential_matcher(embc["z_inv"][:h], embc["z_inv"][h:])
            more.kabsch_transformation_estimation(embc["z_so3"][:h] + embc["t"][:h], embc["z_so3"][h:] + embc["t"][h:])
        tc = time.perf_counter() - t1
        cpu = dict(value=nb / tc, unit="object-instances/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"{nb} instances x {N} pts in one batch: oracle Shape_Prior.encode + sequential_matcher "
                          f"({h}x{h}) + Kabsch ({h}), {tc:.2f} s wall, torch {torch.__version__} CPU")

    if rank == 0:
        total_objects = B * args.steps * world
        line = {
            "metric": "object-instances/sec (encode+match+register), N=1024 pts",
            "value": total_objects / dt,
            "unit": "object-instances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded chair-like clouds; deterministic random-init weights of the released architecture)",
            "config": {"workload": f"BASELINE configs[1]+[2]: batch={B} instances x N={N} pts per GPU = {n_obj}-object scene + rescan; "
                                   f"VN-DGCNN encode, {n_obj}x{n_obj} sequential matching, {n_obj} Kabsch poses",
                       "instances_per_step_per_gpu": B, "points": N, "parallelism": f"instance-sharded x{world}",
                       "steps_in_flight": nfl, "host_enqueue_ms_per_step": round(dt_host / args.steps * 1e3, 3), "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "knn_arithmetic": "canonical (separately rounded mul/add)"},
            "check": {"matches_identity": f"{n_correct}/{n_obj}", "rotations_proper": det_ok,
                      "note": "sanity of the timed work only: weights are untrained (deterministic random init), so the matcher is not expected to recover the identity permutation"},
            "variants": None if dt_fma is None else {
                "knn_fused_multiply_add": {"value": total_objects / dt_fma, "ms_per_step": dt_fma / args.steps * 1e3,
                                           "note": "same steps with LS_FLAG_CONTRACT_FMA (nvcc-style rounding of dist += diff*diff); "
                                                   "bit-exact against the oracle's contract=1 mode; not the headline value"}},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()  # rank 0 may still be in its profiled pass / JSON print
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
