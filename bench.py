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

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from livingscenes_amd import launch  # noqa: E402  (no torch import: `--gpus N` re-executes under the launcher before HIP initialises)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (about 6.3 TB/s achievable)
FP32_PEAK_TFLOPS = 157.3   # fp32 vector == fp32 MFMA peak
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 / f16 MFMA peak (the GEMMs run an fp32 product as three f16 MFMAs -- six bf16 ones under LS_GEMM_MODE=bf16x3: gemm.hip)


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


_PMC = None


def committed_pmc(kernel):
    """PMC figures per launch of operator `kernel` ("kind[layer i]") from the committed per-operator counter passes (scripts/pmc_ops.py under
    rocprofv3 --pmc, one pass per counter group, summarised by scripts/pmc_ops_summary.py; counters cannot be read from inside the process).
    profiles/pmc_latest.json is a POINTER {"see": "<round dir>/pmc/pmc_latest.json"} to the latest round's file (one copy in the tree).
    {} when no committed measurement names the operator."""
    global _PMC
    if _PMC is None:
        root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        try:
            with open(os.path.join(root, "pmc_latest.json")) as f:
                _PMC = json.load(f)
            if "see" in _PMC:
                with open(os.path.join(root, _PMC["see"])) as f:
                    _PMC = json.load(f)
        except (OSError, ValueError):
            _PMC = {}
    return _PMC.get(kernel, {})


def algorithmic_cost(kind, layer, cfg, B, N, bf16x3=True):
    """(bytes, flops, flop peak in TFLOP/s, what the flops are) one launch of operator `kind` at `layer` must move / execute
    (DESIGN.md section 5).  bytes = COMPULSORY HBM bytes: every distinct input row once + the output once (a neighbour gather that
    re-reads a row is served by L2 / Infinity Cache or it is wasted traffic -- it is never counted as useful bytes)."""
    pl = layer_plan(cfg, N)
    L = pl[min(layer, len(pl) - 1)]
    Ns, Nd, Cin, Co = L["Ns"], L["Nd"], L["Cin"], L["Co"]
    f4 = 4
    if not bf16x3:
        mm_peak, mm_what, mm_mult = FP32_PEAK_TFLOPS, "fp32 MFMA flops", 1.0
    elif os.environ.get("LS_GEMM_MODE") == "bf16x3":
        mm_peak, mm_what, mm_mult = BF16_PEAK_TFLOPS, "bf16 MFMA flops as executed (fp32 products = 6 bf16 MFMAs)", 6.0
    else:
        mm_peak, mm_what, mm_mult = BF16_PEAK_TFLOPS, "f16 MFMA flops as executed (an fp32 product = 3 f16 MFMAs: two-piece split)", 3.0
    nc = (10 if L["attn"] else 4) * Co
    pc = (4 if L["attn"] else 2) * Co
    down = Nd != Ns   # neighbour-side columns on the Ns source points, destination-side columns on the Nd selected points
    # attention layers with C_out in {64, 128} compute the destination side inside the edge kernel (edge.hip, edge_attn_fq_kernel): the
    # table holds the neighbour-side columns only; the edge kernel reads the destination points' feature rows + the weights instead
    fused = L["attn"] and bf16x3 and os.environ.get("LS_GEMM_MODE") is None and ((Co == 64 and Cin in (32, 64)) or (Co == 128 and Cin == 64))
    # attention layers with 32 destination points and 128 / 256 input channels (released layers 5, 6): NO table (csrc/edge_fused.hip) -- "gemm_edge" is
    # the operand image of the feature rows (f16 fragment planes), "edge_attn" forms its table slices in LDS on the matrix cores and consumes them there
    fused_t = (L["attn"] and bf16x3 and os.environ.get("LS_GEMM_MODE") is None and Nd == 32
               and ((Cin == 128 and Ns == 128 and down) or (Cin == 256 and Ns == 32 and not down)))
    if fused_t and kind == "gemm_edge":
        rows = B * 3 * (Ns + (Nd if down else 0))
        return 2 * rows * Cin * f4 + rows * 4, 0.0, FP32_PEAK_TFLOPS, "no arithmetic to speak of: feature rows in, (hi, lo) f16 fragment planes out"
    if fused_t and kind == "edge_attn":
        rows = B * 3 * (Ns + (Nd if down else 0))
        flops_tab = mm_mult * 2.0 * B * 3 * (Ns * pc + Nd * (nc - pc)) * Cin           # the same products the table GEMM formed, as executed
        byts = rows * Cin * f4 + nc * Cin * f4 + B * Nd * 16 * 4 + B * Nd * 3 * Co * f4 + 3 * B * (Co // 16) * Nd * 16 * 4   # planes + W + graph + out + scores / norms
        return byts, flops_tab, mm_peak, mm_what + "; table slices formed in LDS (edge_fused.hip: q/k launch, norm sums, soft-max/v launch); the VN activation / score / weighted-sum VALU work (~" + f"{2.0 * B * Nd * 16 * Co * 60 / 1e9:.1f}" + " GFLOP fp32) rides beside it"
    table_floats = B * Ns * 3 * pc if fused else (B * 3 * (Ns * pc + Nd * (nc - pc)) if down else B * Ns * 3 * nc)
    q_side_in = (B * Nd * 3 * Cin + (nc - pc) * Cin) if fused else 0          # floats the fused edge kernel reads instead of Q columns
    q_side_flops = mm_mult * 2.0 * B * Nd * 3 * Cin * (nc - pc) if fused else 0.0
    if kind == "knn":
        D = 3 * Cin
        return (B * ((Nd + Ns) * D * f4 + Nd * 16 * 4), 3.0 * B * Nd * Ns * D, FP32_PEAK_TFLOPS,
                "direct-difference-EQUIVALENT fp32 flops (3 per pair and dimension): the kernels execute fewer -- an f16-MFMA safe filter "
                "over all pairs plus exact canonical distances on the hints and the few survivors (~25 pairs per query)")
    if kind == "gemm_edge":
        return (B * Ns * 3 * Cin * f4 + table_floats * f4 + (pc if fused else nc) * Cin * f4, mm_mult * 2.0 * table_floats * Cin, mm_peak, mm_what)
    if kind in ("edge_attn", "edge_pool"):
        return ((table_floats + q_side_in) * f4 + B * Nd * 16 * 4 + B * Nd * 3 * Co * f4, 2.0 * B * Nd * 16 * Co * (60 if L["attn"] else 30),
                FP32_PEAK_TFLOPS, "fp32 VALU flops (VN activation, scores, soft-max, weighted sum)" +
                (f"; the fused destination-side product adds {q_side_flops / 1e9:.1f} GFLOP of f16 MFMA (microseconds at the matrix peak), not counted here" if fused else ""))
    if kind == "edge_l0":
        return B * Ns * 12 + B * Nd * 16 * 4 + B * Nd * 3 * Co * f4, 2.0 * B * Nd * 16 * Co * 40, FP32_PEAK_TFLOPS, "fp32 VALU flops"
    if kind == "gemm_glob":
        # fused with the VN activation (gemm.hip: gemm_vn_kernel) where C_out % 64 == 0: reads f, writes the activated f' (no [rows, 2C] table)
        glob_fused = Co % 64 == 0 and bf16x3 and os.environ.get("LS_GEMM_MODE") is None
        return (B * Nd * 3 * (2 if glob_fused else 3) * Co * f4 + 4 * Co * Co * f4, mm_mult * 2.0 * B * Nd * 3 * Co * 2 * Co, mm_peak,
                mm_what + (" (+ the VN activation in the epilogue)" if glob_fused else ""))
    if kind == "vn_act":
        return B * Nd * 3 * 3 * Co * f4, 30.0 * B * Nd * Co, FP32_PEAK_TFLOPS, "fp32 VALU flops"
    if kind == "mean":
        # mean over the points + the per-instance half of the global conv's VecLinear in one launch (pointwise.hip: glob_mean_gemv_kernel)
        return (B * Nd * 3 * Co * f4 + 2 * Co * Co * f4 + B * 3 * 2 * Co * f4, 1.0 * B * Nd * 3 * Co + 2.0 * B * 3 * Co * 2 * Co, FP32_PEAK_TFLOPS,
                "fp32 VALU flops (mean over the points + the [3, C] x [C, 2C] contraction of the mean rows)")
    if kind == "fps":
        n = [N] + [p["Nd"] for p in pl if p["Nd"] != p["Ns"]]
        return B * n[min(layer, len(n) - 1)] * 12, 0.0, FP32_PEAK_TFLOPS, "latency-bound (dependent arg-max steps)"
    if kind == "prologue":
        return B * N * 24, 8.0 * B * N * N / 2, FP32_PEAK_TFLOPS, "fp32 VALU flops of the un-pruned pair scan"
    if kind == "gemm_tail":
        return (B * pl[-1]["Nd"] * 3 * (pl[-1]["Co"] + cfg["c_dim"]) * f4, mm_mult * 2.0 * B * pl[-1]["Nd"] * 3 * pl[-1]["Co"] * cfg["c_dim"],
                mm_peak, mm_what)
    if kind == "tail":
        c = cfg["c_dim"]
        return B * pl[-1]["Nd"] * 3 * c * f4 + 2 * c * c * f4, 2.0 * B * 3 * c * c * 2, FP32_PEAK_TFLOPS, "fp32 VALU flops"
    return 0, 0.0, FP32_PEAK_TFLOPS, ""


def roofline_entry(kind, layer, avg_s, cfg, B, N, bf16x3):
    """One roofline object: bound = whichever of (compulsory bytes / HBM peak, flops / pipe peak) is the longer time; `achieved` on
    that axis; `traffic` = PMC HBM bytes per launch (committed counter passes) with the bandwidth they imply."""
    abytes, aflops, fpeak, fwhat = algorithmic_cost(kind, layer, cfg, B, N, bf16x3)
    t_hbm, t_fl = abytes / (HBM_PEAK_GBS * 1e9), aflops / (fpeak * 1e12)
    name = f"{kind}[layer {layer}]"
    if t_hbm >= t_fl:
        e = dict(kernel=name, bound="hbm", achieved=abytes / avg_s / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                 basis="compulsory bytes (distinct input rows once + output once) / measured launch duration")
    else:
        # "mfma" = flops executed on the matrix cores; "valu" = fp32 vector-pipe flops (the VN activation / soft-max / distance arithmetic): both are
        # the contract's compute-bound class, named by the pipe that executes them ("mfma" was hard-coded)
        e = dict(kernel=name, bound="mfma" if fwhat.split(" ")[1:2] == ["MFMA"] else "valu", achieved=aflops / avg_s / 1e12, peak=fpeak, unit="TFLOP/s",
                 basis=fwhat + " / measured launch duration")
    e["frac"] = e["achieved"] / e["peak"]
    if e["bound"] == "valu":
        # the contract's peak (MI355X_MICROARCH.md: 157.3 TFLOP/s vector fp32) counts packed v_pk_fma_f32; round 6 measured that wave64 packed fp32 buys no
        # issue time on this chip (profiles/r6_final/pk_guard_ab.txt: half the instructions, same or longer kernels), so wave64 VALU code tops out at
        # 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz = 78.6 TFLOP/s: `frac` stays on the contract's peak, this is the fraction of the reachable one
        e["frac_of_nonpacked_valu_peak"] = e["achieved"] / (FP32_PEAK_TFLOPS / 2.0)
    e["avg_launch_us"] = avg_s * 1e6
    e["algorithmic_bytes_per_launch"], e["algorithmic_flops_per_launch"] = abytes, aflops
    pmc = committed_pmc(name)
    if "hbm_read_bytes" in pmc and "hbm_write_bytes" in pmc:
        e["traffic"] = pmc["hbm_read_bytes"] + pmc["hbm_write_bytes"]
        e["traffic_over_algorithmic"] = e["traffic"] / max(abytes, 1)
        e["traffic_GBps"] = e["traffic"] / avg_s / 1e9
        e["traffic_frac_of_hbm_peak"] = e["traffic_GBps"] / HBM_PEAK_GBS
        e["traffic_source"] = "profiles/pmc_latest.json -> the latest round's per-operator counter passes (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes, operator run alone)"
    else:
        e["traffic"] = None
    if "mfma_busy_frac" in pmc:
        e["mfma_busy_frac_pmc"] = pmc["mfma_busy_frac"]
    if pmc and "valu_issue_frac" in pmc:   # share of the non-packed VALU issue slots (4 cycles per wave instruction) the operator's kernels use, run alone
        e["valu_issue_frac_pmc"] = pmc["valu_issue_frac"]
        e["wave_wait_frac_pmc"] = pmc.get("wave_wait_frac")
    return e


# NON-PACKED fp32 VALU rate: a wave64 instruction occupies its 16-lane SIMD for 4 cycles -> 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-ops/s
# (the 157.3 TFLOP/s headline = packed v_pk_fma_f32: x2 lanes x2 flops).  Measured, not assumed: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel
# cycles) reaches 0.98 on edge_l0_kernel and 0.6 - 0.8 on the k-NN exact phases (profiles/r4_final/sq_counters_all_kernels.txt); until this
# round's counter sweep this constant was the packed 78.6 T, which made every VALU floor in this file half of what the chip can do.
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9


def knn_hw_utilisation(e, layer, avg_s, cfg, B, N, survivors_per_call, seeded):
    """The k-NN graph build against what the HARDWARE does: `frac` above prices the direct-difference-EQUIVALENT flops of
    SURVEY 8(d), which the kernels do not execute -- a better filter would push it past 1.  frac_hw = the longest of three hardware floors
    over the measured build time:
      hbm   compulsory bytes / 8 TB/s
      mfma  EXECUTED f16 MFMA flops of the safe filter (every (query, candidate) pair once, padded to the 32-wide tiles) / 2.5 PF
      valu  EXECUTED fp32 lane-operations of the exact phases / 39.3 T lane-ops/s (non-packed issue rate): every candidate that gets a canonical distance (the 16 hints
            of a seeded layer + the filter's survivors, counted on the device: ls_profile_knn_stats) costs D subtractions + D multiplications +
            4 D additions -- the canonical chain is serial, and the quad form that keeps the row gathers coalesced executes each add in four
            lanes (csrc/knn_mfma.hip: quad_pair_distance)."""
    pl = layer_plan(cfg, N)
    L = pl[min(layer, len(pl) - 1)]
    D = 3 * L["Cin"]
    pad = lambda n: (n + 31) // 32 * 32
    mfma_flops = 2.0 * B * pad(L["Nd"]) * pad(L["Ns"]) * D
    pairs = B * L["Nd"] * (16 if seeded else 0) + survivors_per_call
    lane_ops = pairs * 6.0 * D
    floors = {"hbm": e["algorithmic_bytes_per_launch"] / (HBM_PEAK_GBS * 1e9), "mfma": mfma_flops / (BF16_PEAK_TFLOPS * 1e12),
              "valu": lane_ops / VALU_LANE_OPS_PER_S}
    which = max(floors, key=floors.get)
    e["frac_hw"] = floors[which] / avg_s
    e["frac_hw_detail"] = {
        "bound": which, "floor_us": {k: round(v * 1e6, 2) for k, v in floors.items()}, "measured_us": round(avg_s * 1e6, 2),
        "executed_f16_mfma_flops": mfma_flops, "executed_exact_pairs": pairs, "exact_pairs_per_query": pairs / (B * L["Nd"]),
        "survivors_per_query": survivors_per_call / (B * L["Nd"]), "executed_valu_lane_ops": lane_ops,
        "useful_valu_lane_ops": pairs * 3.0 * D,
        "note": "frac_hw = max(compulsory bytes / 8 TB/s, executed f16 MFMA flops / 2.5 PFLOP/s, executed exact-phase fp32 lane-ops / 39.3 T lane-ops/s non-packed) "
                "/ measured duration of the whole build (image + seed + sweep + finish launches); `frac` keeps SURVEY 8(d)'s direct-difference-equivalent basis"}
    return e


def gpu_clock_power(index=0, pci=None):
    """{sclk_mhz, mclk_mhz, power_w, source} of one GPU right now (amdgpu sysfs: the `*` line of pp_dpm_sclk / pp_dpm_mclk and hwmon power1_average /
    power1_input in microwatts; `rocm-smi --json` when sysfs is not readable); None when neither answers.  Outside every timed region."""
    import glob
    import re
    import subprocess
    cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
    if pci:
        cards = [d for d in cards if os.path.basename(os.path.realpath(d)).lower() == pci.lower()] or cards
        index = 0 if len(cards) == 1 else index
    try:
        d = cards[index]
        out = {"source": "sysfs"}
        for key, fn in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
            with open(os.path.join(d, fn)) as f:
                cur = [ln for ln in f.read().splitlines() if ln.rstrip().endswith("*")]
            out[key] = int(re.search(r"(\d+)\s*[Mm][Hh]z", cur[0]).group(1)) if cur else None
        for fn in glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input")):
            with open(fn) as f:
                out["power_w"] = round(int(f.read().strip()) / 1e6, 1)
            break
        return out
    except (OSError, IndexError, ValueError, AttributeError):
        pass
    try:
        txt = subprocess.run(["rocm-smi", "-d", str(index), "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=15).stdout
        card = next(iter(json.loads(txt).values()))
        num = lambda v: float(re.search(r"[-+]?\d+(\.\d+)?", str(v)).group(0))
        out = {"source": "rocm-smi"}
        for k, v in card.items():
            kl = k.lower()
            if kl.startswith("sclk clock speed"):
                out["sclk_mhz"] = num(v)
            elif kl.startswith("mclk clock speed"):
                out["mclk_mhz"] = num(v)
            elif "power" in kl and "(w)" in kl and "power_w" not in out:
                out["power_w"] = num(v)
        return out
    except Exception:
        return None


FPS_STEP_FLOOR_US = 0.25   # one dependent arg-max step: an LDS round trip for the winner + a barrier + the DPP wave reduction


def fps_entry(level, avg_s, cfg, B, N):
    """FPS is neither HBM- nor pipe-bound: one workgroup per instance runs n_samples DEPENDENT arg-max steps."""
    n = [N]
    for p in layer_plan(cfg, N):
        if p["Nd"] != p["Ns"]:
            n.append(p["Nd"])
    steps = n[min(level + 1, len(n) - 1)]
    return {"kernel": f"fps[level {level}]", "bound": "latency", "dependent_steps": steps, "avg_launch_us": avg_s * 1e6,
            "ns_per_step": avg_s * 1e9 / steps, "floor_us_per_step": FPS_STEP_FLOOR_US, "achieved": steps / (avg_s * 1e6), "peak": 1.0 / FPS_STEP_FLOOR_US,
            "unit": "dependent steps/us", "frac": FPS_STEP_FLOOR_US * steps / (avg_s * 1e6), "workgroups": B,
            "points_in": n[min(level, len(n) - 1)],
            "basis": f"{steps} dependent arg-max steps x {FPS_STEP_FLOOR_US} us (LDS round trip + workgroup barrier + DPP reduction) / measured launch duration; "
                     f"{B} workgroups = {B / 256:.0%} of the CUs, on a side stream beside layers 0 - 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (one process each); default: the launcher's WORLD_SIZE, 1 without a launcher")
    ap.add_argument("--steps", type=int, default=480,
                    help="timed steps PER BLOCK (a step is ~1.1 ms: one 20-step region is short enough for a single host hiccup or a cold "
                         "allocator to halve the figure -- BENCH_r05 -- hence --blocks)")
    ap.add_argument("--warmup", type=int, default=48, help="untimed steps (raised to >= 3 per in-flight handle and >= --warmup-s of wall time)")
    ap.add_argument("--warmup-s", type=float, default=0.3, help="minimum wall time of the warm-up (clock ramp, every handle's first touches)")
    ap.add_argument("--blocks", type=int, default=0,
                    help="0 = auto (9 when fewer than 48 steps are timed, 3 otherwise).  The K-step timed region (barrier + synchronize on both "
                         "sides, MAX over ranks) is run this many times back to back; `value` / `ms_per_step` are the MEDIAN block, every block's "
                         "time is in config.blocks_ms")
    ap.add_argument("--batch", type=int, default=64, help="instances per step per GPU (two scenes of batch/2 objects)")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--cpu-instances", type=int, default=8, help="bounded sample for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads for the CPU baseline: the reference's op sequence peaks at ~16 threads on the GPU box's "
                         "2x64-core host (0.65 inst/s at 16 vs 0.25 at 128 vs 0.07 at 256; tests/tools/cpu_threads_probe.py)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-fma-variant", action="store_true", help="skip the secondary fused-multiply-add k-NN timing")
    ap.add_argument("--inflight", type=int, default=0,
                    help="0 = auto (see below).  Independent steps kept in flight on separate HIP streams (each with its own model handle and "
                         "workspace): one step's low-occupancy kernels (FPS, heads, matcher) overlap another's big ones.  "
                         "Measured with 16 hardware queues (round 2, after the fusions): 4 -> 37.4k, 8 -> 43.6k, 10 -> 44.6k, "
                         "12 -> 45.6 - 46.4k, 14 -> 45.5k, 16 -> 41.3k obj/s; more hardware queues are worse (12 in flight: 20 queues "
                         "44.8k, 24: 42.6k, 32: 34.6k)")
    args = ap.parse_args()
    if args.inflight <= 0:
        # 12 in the steady state (480 steps: 8 -> 45.9k, 12 -> 47.5k in round 3; round 6: 12 -> 58.3 - 59.8k, 16 -> 53.8k).
        # SHORT runs (the driver's 20 steps): 8.  Round 6 re-measured the 20-step protocol (median of 9 blocks, profiles/r6_final/
        # bench_protocol_depths.txt): 6 / 8 / 10 / 12 in flight = 45.5 / 55.4 - 55.7 / 53.5 - 53.8 / 55.6 - 55.8k -- 8 and 12 are equal, and 8 is the depth
        # the driver measured in rounds 1 - 4 without incident (37.9 -> 49.9k), while its one run at 12 (BENCH_r05) came back at 25.2k, a figure six
        # repetitions of that exact protocol on fresh leases could not reproduce (53.7 - 55.0k, profiles/r6_final/r5_protocol_repro.txt).  What DOES halve a
        # 20-step block is hardware-queue oversubscription (20 in flight on 24 queues: 28.6k, profiles/r6_final/streams_ab.txt): 8 steps x 3 streams on 16
        # queues is the configuration furthest from that cliff at no cost.
        args.inflight = 8 if args.steps < 48 else 12

    # --gpus N is the number of RANKS (one process per GPU).  Started plainly with N > 1 this re-executes itself under
    # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (does not return); started by a launcher it
    # insists on WORLD_SIZE == N -- a `--gpus 8` command can never silently measure one GPU.
    world, rank, local_rank = launch.ensure_ranks(args.gpus, __file__, sys.argv[1:])
    args.gpus = world
    if world > 1:   # one line per rank on stderr, before anything can fail: the launcher's log shows how many ranks really started
        print(f"bench.py: rank {rank} of {world} started (local_rank {local_rank}, pid {os.getpid()})", file=sys.stderr, flush=True)
    global torch
    import torch
    # LS_BENCH_FORCE_DIST=1: run the collective code path (RCCL init, weight broadcast, barriers, all-reduce / all-gather) even with
    # ONE rank -- the only way to exercise the "nccl" backend on the single-GPU test box
    multi = world > 1 or bool(os.environ.get("LS_BENCH_FORCE_DIST"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the hot path)")
    # LS_BENCH_BACKEND=gloo: dry run of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices; the
    # collectives go through the host) -- a logic check only, never a measurement
    backend = os.environ.get("LS_BENCH_BACKEND", "nccl")
    launcher_local = local_rank          # (the shared-device dry run folds local_rank onto the devices that exist; the CPU binding keeps the launcher's)
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"--gpus {args.gpus}: rank {rank} needs device {local_rank} but this node exposes {torch.cuda.device_count()} "
                         f"(RCCL wants one GPU per rank; LS_BENCH_BACKEND=gloo is the shared-device dry run)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one rank per GPU: each rank on the cores next to ITS GPU, disjoint from the other ranks' (livingscenes_amd/launch.py: bind_rank); only a rank
    # that runs the CPU leg (world == 1) keeps a thread pool
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    try:
        props = [torch.cuda.get_device_properties(r % torch.cuda.device_count()) for r in range(n_local)]
        pci = ["%04x:%02x:%02x.0" % (p_.pci_domain_id, p_.pci_bus_id, p_.pci_device_id) for p_ in props]
    except Exception:
        pci = None
    my_cpus = launch.bind_rank(launcher_local if n_local > 1 else 0, n_local, pci)
    if world > 1:
        torch.set_num_threads(1)
        print(f"bench.py: rank {rank} bound to CPUs {launch.format_cpulist(my_cpus)} (device {local_rank})", file=sys.stderr, flush=True)
    import torch.distributed as dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # LS_BENCH_FORCE_DIST without a launcher
            for k, v in (("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(k, v)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from livingscenes_amd import ops, sharding as parallel, synth
    from livingscenes_amd.lib_more.matcher_new import sequential_matcher
    from livingscenes_amd.model_utils import Shape_Prior

    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    if rank == 0:
        ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    else:  # other ranks start from zeros and receive rank 0's weights over RCCL
        ew = {k: torch.zeros(s) for k, s in synth.encoder_param_shapes(ecfg).items()}
        dw = {k: torch.zeros_like(v) for k, v in synth.make_decoder_weights(dcfg, 0).items()}
    nfl = max(1, args.inflight)
    if nfl > 1:
        # A/B on MI355X (scripts in DESIGN.md 6): intra-step GEMM||k-NN stream overlap is +5 % for a single in-flight step but
        # -3.5 % once two whole steps already overlap; the library default stays on, the bench turns it off on every handle
        # (ls_model_set_option(LS_OPT_GEMM_OVERLAP, 0), below)
        pass
    sp = Shape_Prior.from_state(ecfg, dcfg, ew, dw, device=dev)
    if multi:
        parallel.broadcast_weights(sp, src=0)
    # one model handle (packed weights + side stream + workspace) per in-flight step; weights are shared tensors
    sps = [sp] + [Shape_Prior.from_state(ecfg, dcfg, sp.encoder.state_dict(), sp.decoder.F.state_dict(), device=dev) for _ in range(nfl - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    if nfl > 1:
        from livingscenes_amd import _lib as _ls_lib
        for s_ in sps:
            s_.hip_model().set_option(_ls_lib.OPT_GEMM_OVERLAP, 0)

    B, N = args.batch, args.points
    n_obj = B // 2
    scene = synth.make_scene_pair(n_obj, N, seed=1000 + rank)
    x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)  # [B,3,N] resident in HBM

    def step(sp=sp):
        emb = sp.encode(x)
        m = sequential_matcher(emb["z_inv"][:n_obj], emb["z_inv"][n_obj:])
        # kabsch_transformation_estimation(z_so3 + t of the reference objects, z_so3 + t of their matches) (more_solver.py:114-116): the
        # sums, the gather by matches0 and its clamp happen inside the Kabsch launch (ls_kabsch_codes_f32), no ATen kernel in the step
        R, t = ops.kabsch_codes(emb["z_so3"][:n_obj], emb["t"][:n_obj], emb["z_so3"][n_obj:], emb["t"][n_obj:], sel2=m["matches0"])
        return emb, m, R, t

    def barrier():
        if multi:
            dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()

    last = [None] * nfl    # every handle's most recent result (the post-run check compares ALL of them, not one)
    # (Round 4: enqueuing the in-flight steps from several host threads -- one per stream, ctypes releases the GIL inside the library call -- was
    #  measured and NOT kept: the HIP launch path serialises, host time per step 0.30 -> 0.55 - 0.65 ms with 4 - 8 threads, the 20-step figure 45.7k -> 44.2k.)

    def run(n, events=None):
        out = None
        for i in range(n):
            with torch.cuda.stream(streams[i % nfl]):
                out = last[i % nfl] = step(sps[i % nfl])
                if events is not None:
                    events[i].record()
        return out

    # PROTOCOL (round 6, after BENCH_r05 = 25.2k against 53.5k on the same command here -- VERDICT r5 item 1):
    #   warm-up   >= 3 steps on EVERY in-flight handle (the 2nd step of a handle still holds the 1st one's outputs -> the caching allocator grows;
    #             `hipMalloc` inside the timed region was the r5 hazard: its warm-up was ONE step per handle) and >= --warmup-s of wall time
    #             (clock ramp from idle), in synchronised rounds of one step per handle;
    #   blocks    the contract's timed region -- barrier + synchronize, EXACTLY K steps, synchronize + barrier -- R times back to back; per block
    #             the MAX over ranks; `value` / `ms_per_step` = the MEDIAN block.  One slow block (a host hiccup is ~10 ms, a 20-step block ~25 ms)
    #             moves the median by nothing; every block's time is reported in `config.blocks_ms` (the driver keeps `config`).
    n_blocks = args.blocks if args.blocks > 0 else (9 if args.steps < 48 else 3)
    r5_protocol = bool(os.environ.get("LS_BENCH_R5_PROTOCOL"))   # dev: round 5's region (ONE step of warm-up per handle, one block) to reproduce BENCH_r05
    if r5_protocol:
        n_blocks, args.warmup_s = 1, 0.0
    my_pci = pci[launcher_local % len(pci)] if pci else None
    clk_idle = gpu_clock_power(local_rank, my_pci) if rank == 0 else None
    with torch.no_grad():
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        t_w = time.perf_counter()
        n_warm = max(args.warmup, nfl if r5_protocol else 3 * nfl)
        out = run(n_warm)
        torch.cuda.synchronize()
        while time.perf_counter() - t_w < args.warmup_s:
            out = run(nfl)
            n_warm += nfl
            torch.cuda.synchronize()
        warm_s = time.perf_counter() - t_w
        clk_before = gpu_clock_power(local_rank, my_pci) if rank == 0 else None
        blocks = []      # (dt, dt_host, sorted completion times)
        for _ in range(n_blocks):
            barrier()
            torch.cuda.synchronize()
            step_done = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
            ev0 = torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()
            out = run(args.steps, step_done)
            dt_host_b = time.perf_counter() - t0     # host time to enqueue the K steps (no device sync inside a step)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            dt_b = time.perf_counter() - t0
            blocks.append((dt_b, dt_host_b, sorted(ev0.elapsed_time(e) for e in step_done)))    # numbers only: no block's tensors are retained
    clk_after = gpu_clock_power(local_rank, my_pci) if rank == 0 else None
    # MAX over ranks per block, then the median block (by the all-rank time, so that every rank picks the same one)
    blk_t = torch.tensor([b_[0] for b_ in blocks], device=dev, dtype=torch.float64)
    if multi:
        dist.all_reduce(blk_t, op=dist.ReduceOp.MAX)
    blk_all = [float(v) for v in blk_t.cpu()]
    med_i = sorted(range(n_blocks), key=lambda i: blk_all[i])[(n_blocks - 1) // 2]
    dt, dt_host, done_ms = blocks[med_i]    # this rank's figures of the median block
    last_main = list(last)                  # every handle's result of the LAST block (checked below, outside the timed regions; the FMA run overwrites `last`)
    # per-step completion times (event per step on its stream): the spread of the inter-completion intervals shows a host hiccup or
    # a straggling step that the K-step mean hides
    def gap_stats(done):
        gaps = sorted(b - a for a, b in zip([0.0] + done[:-1], done))
        return {"inter_completion_ms_min": round(gaps[0], 4), "inter_completion_ms_median": round(gaps[len(gaps) // 2], 4),
                "inter_completion_ms_p90": round(gaps[int(0.9 * (len(gaps) - 1))], 4), "inter_completion_ms_max": round(gaps[-1], 4),
                "last_step_done_ms": round(done[-1], 3)}
    if os.environ.get("LS_BENCH_DUMP_STEPS") and rank == 0:      # dev: the completion time of every timed step (ramp / drain shape)
        for bi, b_ in enumerate(blocks):
            print(f"block {bi} step completion ms:", " ".join(f"{v:.2f}" for v in b_[2]), f"| host enqueue {b_[1] * 1e3:.2f} ms | total {b_[0] * 1e3:.2f} ms",
                  file=sys.stderr)
    step_stats = gap_stats(done_ms)
    blocks_ms = [round(v * 1e3, 3) for v in blk_all]
    block_stats = {"blocks": n_blocks, "blocks_ms": blocks_ms, "block_ms_min": min(blocks_ms), "block_ms_median": round(blk_all[med_i] * 1e3, 3),
                   "block_ms_max": max(blocks_ms), "block_max_over_min": round(max(blocks_ms) / min(blocks_ms), 4),
                   "inter_completion_ms_max_per_block": [gap_stats(b_[2])["inter_completion_ms_max"] for b_ in blocks],
                   "host_enqueue_ms_per_block": [round(b_[1] * 1e3, 3) for b_ in blocks],
                   "warmup_steps_run": n_warm, "warmup_wall_s": round(warm_s, 3),
                   "gpu_clock_power_idle": clk_idle, "gpu_clock_power_after_warmup": clk_before, "gpu_clock_power_after_blocks": clk_after}
    # (+ the rank's CPU set as up to four 64-bit masks -> 256 hardware threads, so that a host-bound or doubly-booked rank shows in the line)
    cpu_masks = [float(sum(1 << (c - 52 * k) for c in my_cpus if 52 * k <= c < 52 * (k + 1))) for k in range(8)]
    my = torch.tensor([dt, dt_host, float(torch.cuda.current_device()), float(B * args.steps)] + cpu_masks, device=dev, dtype=torch.float64)
    dt_t = my[:1].clone()
    per_rank = None
    if multi:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
        mine = my if backend == "nccl" else my.cpu()     # gloo gathers through host memory only
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        def cpus_of(v):
            return launch.format_cpulist([52 * k + b for k in range(8) for b in range(52) if (int(v[4 + k]) >> b) & 1])
        per_rank = [{"rank": r, "device": int(v[2]), "objects": int(v[3]), "ms_per_step": round(float(v[0]) / args.steps * 1e3, 4),
                     "host_enqueue_ms_per_step": round(float(v[1]) / args.steps * 1e3, 4), "cpus": cpus_of(v)} for r, v in enumerate(allr)]
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
        if multi:
            dist.all_reduce(dt_f, op=dist.ReduceOp.MAX)
        dt_fma = float(dt_f.item())

    # the un-profiled ONE-step-in-flight time (single stream, handle 0; the library's own side streams stay on): median of 3 x 20 steps.
    # Beside ms_per_step it says how much of the figure is the overlap of whole steps and how much is the single step.
    one_ms = []
    with torch.no_grad():
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                for _i in range(20):
                    step(sps[0])
            torch.cuda.synchronize()
            one_ms.append((time.perf_counter() - t1) / 20 * 1e3)
    one_in_flight_ms = sorted(one_ms)[1]

    emb, m, R, t = out
    # every handle of the timed region saw the same batch: their last results must agree BIT FOR BIT (reproducibility with all the
    # streams in flight); checked outside the timed region on every handle that ran
    def same(a, b):
        return all(torch.equal(a[0][k], b[0][k]) for k in ("z_so3", "z_inv", "s", "t")) and torch.equal(a[1]["matches0"], b[1]["matches0"]) \
            and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    ran = [r for r in last_main if r is not None]
    handles_identical = all(same(ran[0], r) for r in ran[1:])
    # sanity of the measured work (outside the timed region): matches are the identity permutation, poses are rotations
    n_correct = int((m["matches0"].cpu() == torch.arange(n_obj)).sum())
    det_ok = bool((torch.det(R.cpu()) > 0.99).all())
    if multi:  # RCCL all-gather of the per-rank codes (4.1 KB each: latency-bound; outside the timed region)
        allc = parallel.all_gather_codes(emb)
        assert allc["z_inv"].shape[0] == B * world

    roof = None
    bf16x3 = os.environ.get("LS_GEMM_MODE") != "fp32"      # (False: exact fp32 MFMA chains, LS_GEMM_MODE=fp32)
    if rank == 0 and not args.no_profile:
        hip = sp.hip_model()
        prof_steps = min(args.steps, 24)      # per-launch hipEvent pairs: a bounded, serial pass on one stream
        hip.profile_begin()
        with torch.no_grad():
            for _ in range(prof_steps):
                step()
        knn_stats = hip.profile_knn_stats()     # {layer: (candidates given a canonical distance beyond the hints, queries)} over the profiled steps
        prof = hip.profile_end()
        tot = sum(p["total_ms"] for p in prof)
        by = sorted(prof, key=lambda p: -p["total_ms"])
        # dominant kernel = the largest launch of the operator family with the largest share of device time
        fam = {}
        for q in prof:
            fam[q["kind"]] = fam.get(q["kind"], 0.0) + q["total_ms"]
        dom_kind = max(fam, key=fam.get)
        # ... and of its launches within 3 % of the family's longest, the one FURTHEST below its roofline (layers 2 and 3 of the attention family
        # are within a microsecond or two of each other: picking by time alone flipped between them from run to run)
        dom_c = [q for q in prof if q["kind"] == dom_kind]
        dom_t = max(q["total_ms"] / q["launches"] for q in dom_c)
        dom = min((q for q in dom_c if q["total_ms"] / q["launches"] >= 0.97 * dom_t),
                  key=lambda q: roofline_entry(q["kind"], q["layer"], q["total_ms"] / q["launches"] * 1e-3, ecfg, B, N, bf16x3)["frac"])
        roof = roofline_entry(dom["kind"], dom["layer"], dom["total_ms"] / dom["launches"] * 1e-3, ecfg, B, N, bf16x3)
        if dom["kind"] == "knn":
            roof["note"] = ("one k-NN graph build = the launch sequence of that layer (seeded layers 1 / 2: f16 image incl. centre, seed, sweep, finish = 4 - 5 launches; un-seeded layers 3 / 4: image, sweep, finish = 3); "
                            "bound by fp32 VALU issue on the direct-difference-equivalent count -- see `basis`")
        # the fused k-NN kernel (knn_mfma.hip: knn_fused_kernel, round 5) uses no hints: every exact distance is a survivor's, counted on the device
        seeded_layers = set()

        def with_hw(e, q):
            if q["kind"] == "knn" and q["layer"] in knn_stats and knn_stats[q["layer"]][1]:
                knn_hw_utilisation(e, q["layer"], q["total_ms"] / q["launches"] * 1e-3, ecfg, B, N, knn_stats[q["layer"]][0] / max(q["launches"], 1),
                                   q["layer"] in seeded_layers)
            return e
        with_hw(roof, dom)
        roof["timing"] = f"hipEvent pair per launch on the launching stream, separate profiled pass of {prof_steps} steps (one step in flight)"
        roof["share_of_device_time"] = dom["total_ms"] / max(tot, 1e-9)
        roof["breakdown_ms_per_step"] = {k: round(v / prof_steps, 4) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
        roof["per_layer_ms_per_step"] = {f"{p['kind']}{p['layer']}": round(p["total_ms"] / prof_steps, 4) for p in by[:40]}
        # the other operator families north_star names, on their largest launch: the VN edge-conv gather kernel (HBM-bound:
        # target >= 30 % of the HBM roofline) and the VN-Linear table GEMM (write-bound at small K, matrix-core-bound at large K)
        extra = []
        for kind in ("edge_attn", "edge_pool", "gemm_edge", "gemm_glob"):
            cands = [p for p in prof if p["kind"] == kind]
            if not cands:
                continue
            e = max(cands, key=lambda p: p["total_ms"])
            extra.append(roofline_entry(kind, e["layer"], e["total_ms"] / e["launches"] * 1e-3, ecfg, B, N, bf16x3))
        for e in sorted((p for p in prof if p["kind"] == "edge_attn" and 2 <= p["layer"] <= 4), key=lambda p: p["layer"]):   # every fused-gather layer
            if all(x["kernel"] != f"edge_attn[layer {e['layer']}]" for x in extra):
                extra.append(roofline_entry("edge_attn", e["layer"], e["total_ms"] / e["launches"] * 1e-3, ecfg, B, N, bf16x3))
        if cands := [p for p in prof if p["kind"] == "gemm_edge"]:   # and the most matrix-core-heavy table GEMM (largest K)
            e = max(cands, key=lambda p: p["layer"])
            if all(x["kernel"] != f"gemm_edge[layer {e['layer']}]" for x in extra):
                extra.append(roofline_entry("gemm_edge", e["layer"], e["total_ms"] / e["launches"] * 1e-3, ecfg, B, N, bf16x3))
        for q in sorted((p for p in prof if p["kind"] == "knn" and p["layer"] in knn_stats and p is not dom), key=lambda p: p["layer"]):   # every filter-path k-NN layer
            extra.append(with_hw(roofline_entry("knn", q["layer"], q["total_ms"] / q["launches"] * 1e-3, ecfg, B, N, bf16x3), q))
        for q in sorted((p for p in prof if p["kind"] == "fps"), key=lambda p: p["layer"])[:1]:     # the single heaviest kernel of the step: FPS level 0
            extra.append(fps_entry(q["layer"], q["total_ms"] / q["launches"] * 1e-3, ecfg, B, N))
        roof["other_kernels"] = extra
        # the whole step against its own floors: every profiled operator's compulsory bytes and max(bytes / HBM peak, flops / pipe peak)
        ws_bytes = ws_floor = ws_floor_hw = ws_pmc = ws_valu = 0.0
        pmc_missing = []
        for q in prof:
            per = q["launches"] / prof_steps
            ab, af, fpk, _ = algorithmic_cost(q["kind"], q["layer"], ecfg, B, N, bf16x3)
            fl = max(ab / (HBM_PEAK_GBS * 1e9), af / (fpk * 1e12))
            ws_bytes += ab * per
            ws_floor += fl * per
            hw = fl
            if q["kind"] == "knn" and q["layer"] in knn_stats and knn_stats[q["layer"]][1]:
                tmp = knn_hw_utilisation({"algorithmic_bytes_per_launch": ab}, q["layer"], 1.0, ecfg, B, N, knn_stats[q["layer"]][0] / max(q["launches"], 1),
                                         q["layer"] in seeded_layers)
                hw = tmp["frac_hw"]            # (avg_s = 1: frac_hw is the floor in seconds)
            elif q["kind"] == "fps":
                hw = fps_entry(q["layer"], 1.0, ecfg, B, N)["frac"]      # (avg_s = 1: the floor in seconds)
            ws_floor_hw += hw * per
            # the counter passes run the residual global conv (mean + GEMM launches) and the heads (conv_c GEMM + tail kernel) as ONE operator each:
            # their bytes are booked on the GEMM row, the companion row adds nothing
            if q["kind"] == "gemm_glob":
                pm, calls = committed_pmc(f"global_conv[layer {q['layer']}]"), 1.0
            elif q["kind"] == "gemm_tail":
                pm, calls = committed_pmc("tail[layer 0]"), 1.0
            elif q["kind"] == "mean":
                pm, calls = ({"hbm_read_bytes": 0, "hbm_write_bytes": 0} if committed_pmc(f"global_conv[layer {q['layer']}]") else {}), 1.0
            elif q["kind"] == "tail":
                pm, calls = ({"hbm_read_bytes": 0, "hbm_write_bytes": 0} if committed_pmc("tail[layer 0]") else {}), 1.0
            else:
                pm, calls = committed_pmc(f"{q['kind']}[layer {q['layer']}]"), per
            ws_valu += pm.get("valu_wave_insts", 0.0) * calls
            if "hbm_read_bytes" in pm and "hbm_write_bytes" in pm:
                ws_pmc += (pm["hbm_read_bytes"] + pm["hbm_write_bytes"]) * calls
            else:
                pmc_missing.append(f"{q['kind']}[{q['layer']}]")
        roof["whole_step"] = {
            "algorithmic_bytes": ws_bytes, "pmc_bytes": ws_pmc or None, "pmc_over_algorithmic": (ws_pmc / ws_bytes) if ws_pmc else None,
            "pmc_operators_without_a_committed_counter_pass": pmc_missing,
            "sum_floor_ms": ws_floor * 1e3, "sum_floor_hw_ms": ws_floor_hw * 1e3, "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_one_in_flight_profiled": tot / prof_steps,
            "frac_of_sum_floor": ws_floor / (dt / args.steps), "frac_of_sum_floor_hw": ws_floor_hw / (dt / args.steps),
            "algorithmic_GBps": ws_bytes / (dt / args.steps) / 1e9, "algorithmic_frac_of_hbm_peak": ws_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            # executed VALU work (committed SQ_INSTS_VALU passes): wave-instructions x 64 lanes against the non-packed issue rate of the chip
            "valu_wave_insts_pmc": ws_valu or None, "valu_issue_ms_pmc": (ws_valu * 64.0 / VALU_LANE_OPS_PER_S * 1e3) if ws_valu else None,
            "valu_issue_frac_of_step": (ws_valu * 64.0 / VALU_LANE_OPS_PER_S / (dt / args.steps)) if ws_valu else None,
            "note": "sum over every launch of one step of max(compulsory bytes / 8 TB/s, flops / pipe peak) (sum_floor_ms: k-NN on the direct-difference-equivalent "
                    "basis; sum_floor_hw_ms: k-NN on executed work, FPS at 0.25 us per dependent step) against the timed ms_per_step; floors of kernels "
                    "that run concurrently on different streams are still summed, so this is a lower bound on a serial schedule, not on the chip; "
                    "valu_issue_ms_pmc: the VALU instructions the step's kernels EXECUTE (counter passes, 4 issue cycles each) -- the part of the step that only "
                    "fewer instructions can shorten"}

    cpu = None
    oracle_check = None
    if rank == 0 and world == 1 and args.cpu_instances > 0:
        from oracle import more, net  # the checker, timed as the CPU baseline ("port" of the reference's op sequence)
        import statistics
        nb = args.cpu_instances
        torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
        ewc = synth.make_encoder_weights(ecfg, 0)

        def cpu_pass(xc):
            t1 = time.perf_counter()
            with torch.no_grad():
                embc = net.shape_prior_encode(ewc, ecfg, xc)
                h = max(1, xc.shape[0] // 2)
                if xc.shape[0] >= 2:
                    more.sequential_matcher(embc["z_inv"][:h], embc["z_inv"][h:2 * h])
                    more.kabsch_transformation_estimation(embc["z_so3"][:h] + embc["t"][:h], embc["z_so3"][h:2 * h] + embc["t"][h:2 * h])
            return time.perf_counter() - t1, embc
        # warm-up 1, median of >= 3 runs, B = 1 and B = nb (instances spread over the batch: rows 0, 9, 18, ...)
        sel = [(i * (B - 1)) // max(nb - 1, 1) for i in range(nb)]
        xc1, xcn = x[:1].cpu(), x[sel].cpu()
        cpu_pass(xc1)
        t_b1 = statistics.median(cpu_pass(xc1)[0] for _ in range(3))
        runs = [cpu_pass(xcn) for _ in range(3)]
        t_bn = statistics.median(r[0] for r in runs)
        cpu = dict(value=nb / t_bn, unit="object-instances/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"median of 3 passes (after 1 warm-up) over {nb} instances x {N} pts in one batch: oracle Shape_Prior.encode + "
                          f"sequential_matcher ({nb // 2}x{nb // 2}) + Kabsch ({nb // 2}); {t_bn:.2f} s per pass; B = 1: {1.0 / t_b1:.3f} instances/s "
                          f"({t_b1:.2f} s per pass); torch {torch.__version__} CPU, {torch.get_num_threads()} threads of {os.cpu_count()} logical cores")
        # correctness of the TIMED work against the oracle (outside the timed region, on the cpu_baseline leg): the same nb instances
        embc = runs[-1][1]

        def rel(a, b_):
            return float((a.double().cpu() - b_.double()).abs().max() / b_.double().abs().max().clamp_min(1e-30))
        oracle_check = {k: rel(emb[k][sel], embc[k]) for k in ("z_so3", "z_inv", "s", "t")}
        oracle_check["instances"] = sel
        oracle_check["tolerance"] = 1e-4
        oracle_check["ok"] = all(v < 1e-4 for k, v in oracle_check.items() if k in ("z_so3", "z_inv", "s", "t"))
        ref_m = more.sequential_matcher(emb["z_inv"][:n_obj].cpu(), emb["z_inv"][n_obj:].cpu())
        oracle_check["matches_bit_exact_vs_oracle_on_hip_codes"] = bool(torch.equal(ref_m["matches0"], m["matches0"].cpu()))

    line = None
    if rank == 0:
        total_objects = B * args.steps * world
        line = {
            "metric": "object-instances/sec (encode+match+register), N=1024 pts",
            "value": total_objects / dt,
            "unit": "object-instances/s",
            "n_gpus": world,
            "ranks": world,
            "collective_backend": (("rccl" if backend == "nccl" else backend + " (shared-device dry run: NOT a measurement)") if multi else None),
            "rccl_ranks": world if (multi and backend == "nccl") else (1 if not multi else 0),
            "devices": [p_["device"] for p_ in per_rank] if per_rank else [torch.cuda.current_device()],
            "device_name": torch.cuda.get_device_name(dev),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" + ((" (GEMM products as three-piece bf16 splits on the bf16 matrix cores, fp32 accumulate: as accurate as an fp32 FMA chain)"
                              if os.environ.get("LS_GEMM_MODE") == "bf16x3" else
                              " (GEMM products as two-piece f16 splits on the f16 matrix cores, fp32 accumulate: as accurate against fp64 as an fp32 FMA chain)")
                             if bf16x3 else ""),
            "data": "synthetic (seeded chair-like clouds; deterministic random-init weights of the released architecture)",
            "config": {"workload": f"BASELINE configs[1]+[2]: batch={B} instances x N={N} pts per GPU = {n_obj}-object scene + rescan; "
                                   f"VN-DGCNN encode, {n_obj}x{n_obj} sequential matching, {n_obj} Kabsch poses",
                       "instances_per_step_per_gpu": B, "points": N, "parallelism": f"instance-sharded x{world}",
                       "steps_in_flight": nfl, "protocol": f"median of {n_blocks} back-to-back blocks of {args.steps} steps (each: barrier + synchronize, K steps, "
                                                            f"synchronize + barrier; MAX over ranks per block) after {n_warm} warm-up steps / {warm_s:.2f} s",
                       **block_stats, **{"median_block_" + k: v for k, v in step_stats.items()},
                       "ms_per_step_one_in_flight_unprofiled": round(one_in_flight_ms, 4),
                       "host_enqueue_ms_per_step": round(max([p_["host_enqueue_ms_per_step"] for p_ in per_rank] if per_rank else [dt_host / args.steps * 1e3]), 3),
                       "host_enqueue_basis": "max over ranks" if per_rank else "this rank", "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "knn_arithmetic": "canonical (separately rounded mul/add)"},
            "check": {"oracle_relerr": oracle_check, "rotations_proper": det_ok, "matches_identity": f"{n_correct}/{n_obj}",
                      "handles_bit_identical": f"{handles_identical} ({len(ran)} handles, last step of each)",
                      "note": "oracle_relerr: max-norm relative error of the timed codes against the CPU oracle on instances spread over the batch "
                              "(tests/test_hip_fullbatch.py checks the same batch in pytest); weights are untrained, so the matcher is not expected "
                              "to recover the identity permutation"},
            "step_stats": step_stats,
            "per_rank": per_rank,
            "variants": None if dt_fma is None else {
                "knn_fused_multiply_add": {"value": total_objects / dt_fma, "ms_per_step": dt_fma / args.steps * 1e3,
                                           "note": "same steps with LS_FLAG_CONTRACT_FMA (nvcc-style rounding of dist += diff*diff); "
                                                   "bit-exact against the oracle's contract=1 mode; not the headline value"}},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
    if multi:
        dist.barrier(device_ids=[local_rank]) if backend == "nccl" else dist.barrier()  # rank 0 may still be in its profiled pass
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints its version banner through C stdio (flushed at exit when piped)
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
