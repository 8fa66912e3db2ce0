"""BASELINE configs[1]+[2] at FULL batch, exactly as bench.py runs it: 64 instances x 1024 points (the 32-object scene and its
rescan, seed 1000), 12 model handles on 12 streams all in flight (the configuration bench.py times), GPU_MAX_HW_QUEUES=16.  Kernel dispatch depends on B (candidate
split heuristics, un-split grids, XCD remap, handles sharing the chip), so the B <= 3 encoder parity cases of
test_hip_parity.py do not cover this configuration.

Checks (reference: model_utils.py:165-197, lib_more/matcher_new.py:109-139, lib_more/pose_estimation.py:29-102):
  * 8 instances spread over the batch (rows 0, 9, ..., 63) against oracle.net.shape_prior_encode: z_so3 / z_inv / s / t within
    1e-4 of max-norm, FPS and layer-0 k-NN indices exact, per-layer k-NN agreement > 99.5 %;
  * all 12 handles return bit-identical codes, matches and poses (and equal to a later single traced pass);
  * sequential_matcher assignments BIT-EXACT and Kabsch poses within 1e-4 against oracle.more run on the HIP codes.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from livingscenes_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def fullbatch(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fullbatch") / "out.npz")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    subprocess.run([sys.executable, os.path.join(REPO, "tests", "tools", "fullbatch_worker.py"), out, "32", "1024", "12"],
                   check=True, env=env, cwd=REPO, timeout=900)
    return dict(np.load(out))


def test_fullbatch_handles_are_bit_identical(fullbatch):
    r = fullbatch
    assert int(r["hw_queues"]) == 16 and int(r["handles"]) == 12
    for i in range(1, 12):
        for k in ("z_so3", "z_inv", "s", "t", "m0", "m1", "pose_R", "pose_t"):
            assert np.array_equal(r[f"h{i}_{k}"], r[f"h0_{k}"]), (i, k)
    for k in ("z_so3", "z_inv", "s"):
        assert np.array_equal(r[f"tr_{k}"], r[f"h0_{k}"]), k
    assert np.array_equal(r["tr_t"], r["h0_t"].reshape(64, 3))


def test_fullbatch_encode_vs_oracle(fullbatch):
    from oracle import net
    r = fullbatch
    ecfg = synth.default_encoder_cfg()
    ew = synth.make_encoder_weights(ecfg, 0)
    scene = synth.make_scene_pair(32, 1024, seed=1000)
    x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous()
    assert np.array_equal(x.numpy(), r["x"])                 # the worker ran the bench batch
    sel = [0, 9, 18, 27, 36, 45, 54, 63]
    tr = {}
    ref = net.shape_prior_encode(ew, ecfg, x[sel], trace=tr)
    for k in ("z_so3", "z_inv", "s", "t"):
        assert relerr(r[f"h0_{k}"][sel], ref[k].numpy()) < TOL, k
    assert np.array_equal(r["knn_0"][sel], tr["knn_idx_0"].numpy().astype(np.int32))
    for j, i in enumerate(ecfg["down_sample_layers"]):
        assert np.array_equal(r[f"fps_{j}"][sel], tr[f"fps_idx_{i}"].numpy().astype(np.int32)), f"fps level {j}"
    for i in range(1, ecfg["num_layers"]):
        rate = (r[f"knn_{i}"][sel] == tr[f"knn_idx_{i}"].numpy()).mean()
        assert rate > 0.995, f"layer {i}: k-NN index agreement {rate:.5f}"


def test_fullbatch_match_and_register_vs_oracle(fullbatch):
    from oracle import more
    r = fullbatch
    n = 32
    z_inv, z_so3, t = torch.from_numpy(r["h0_z_inv"]), torch.from_numpy(r["h0_z_so3"]), torch.from_numpy(r["h0_t"])
    ref = more.sequential_matcher(z_inv[:n], z_inv[n:])
    assert np.array_equal(r["h0_m0"], ref["matches0"].numpy())
    assert np.array_equal(r["h0_m1"], ref["matches1"].numpy())
    j = ref["matches0"].clamp(min=0)
    p1 = z_so3[:n] + t[:n]
    p2 = (z_so3[n:] + t[n:]).index_select(0, j)
    R, tt, _, _ = more.kabsch_transformation_estimation(p1, p2)
    assert relerr(r["h0_pose_R"], R.numpy()) < TOL
    assert relerr(r["h0_pose_t"], tt.numpy()) < TOL


def test_encode_graph_replay_is_bit_identical():
    """LS_OPT_ENCODE_GRAPH (opt-in): the captured hipGraph of ls_encode's launch sequence (incl. the fork / join onto the FPS side
    stream), replayed with x and the outputs at NEW addresses every call, returns exactly the codes of the direct path."""
    from livingscenes_amd import _lib
    from livingscenes_amd.model_utils import Shape_Prior
    dev = torch.device("cuda:0")
    ecfg, dcfg = synth.small_encoder_cfg(), synth.small_decoder_cfg()
    sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 1), synth.make_decoder_weights(dcfg, 1), device=dev)
    xs = [synth.make_instances(5, 256, seed=70 + i).to(dev) for i in range(3)]
    st = torch.cuda.Stream(device=dev)
    with torch.no_grad():
        direct = [sp.encode(x) for x in xs]
        torch.cuda.synchronize()
        sp.hip_model().set_option(_lib.OPT_ENCODE_GRAPH, 1)
        with torch.cuda.stream(st):
            for rep in range(2):
                keep = []
                for x, want in zip(xs, direct):
                    got = sp.encode(x.clone())          # a fresh input address every call
                    keep.append(got)
                st.synchronize()
                for got, want in zip(keep, direct):
                    for k in ("z_so3", "z_inv", "s", "t"):
                        assert torch.equal(got[k], want[k]), (rep, k)
        sp.hip_model().set_option(_lib.OPT_ENCODE_GRAPH, 0)


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _codes_sha(sp, B, seed):
    import hashlib
    x = synth.make_instances(B, 1024, seed=seed)
    x = (x if isinstance(x, torch.Tensor) else x[0]).to(_dev())
    with torch.no_grad():
        c = sp.encode(x)
    return hashlib.sha1(b"".join(c[k].cpu().numpy().tobytes() for k in ("z_so3", "z_inv", "s", "t"))).hexdigest()


def test_fused_destination_side_equals_table_path_bit_for_bit():
    """Attention layers 2 - 4 compute the destination-side column groups inside the edge kernel (edge_attn_fq_kernel: f16-split MFMA
    product per workgroup) instead of reading them from the table the GEMM wrote (ls_model_set_option(LS_OPT_EDGE_FUSE_Q, 0)).  Same products,
    same accumulation order, same additions afterwards: the codes of the released-width encoder must be IDENTICAL, ragged batch included."""
    from livingscenes_amd import _lib
    from livingscenes_amd.model_utils import Shape_Prior
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=_dev())
    hip = sp.hip_model()
    outs = []
    for mode in (1, 0):
        prev = hip.set_option(_lib.OPT_EDGE_FUSE_Q, mode)
        outs.append([_codes_sha(sp, B, seed) for B, seed in ((7, 3), (64, 1000))])
        hip.set_option(_lib.OPT_EDGE_FUSE_Q, prev)
    assert outs[0] == outs[1]


def test_fused_global_conv_matches_the_two_launch_path():
    """The residual global conv as one launch (gemm_vn_kernel: the per-point GEMM with the VN activation as its epilogue) against
    GEMM -> table -> vn_act_rows (LS_OPT_GLOB_FUSE = 0): same products in the same order, same activation formula.  Compared through the
    operator export on released widths (layers 2, 4, 6), ragged point counts included."""
    from livingscenes_amd import _lib
    from livingscenes_amd.model_utils import Shape_Prior
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=_dev())
    hip = sp.hip_model()
    g = torch.Generator().manual_seed(5)
    for layer, B, N in ((2, 3, 512), (2, 2, 333), (4, 5, 128), (6, 7, 32), (6, 1, 11)):
        C = ecfg["feat_dim"][layer]
        msg = torch.randn(B, N, 3, C, generator=g).to(_dev())
        a = hip.vn_lna_global(layer, msg).cpu().numpy()
        prev = hip.set_option(_lib.OPT_GLOB_FUSE, 0)
        b = hip.vn_lna_global(layer, msg).cpu().numpy()
        hip.set_option(_lib.OPT_GLOB_FUSE, prev)
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max(), (layer, B, N, np.abs(a - b).max())


def test_streaming_global_conv_equals_the_tiled_kernels_bit_for_bit():
    """Layers 2 / 3 (64 channels) run the residual global conv as gemm_vn_direct_kernel when the producer hands over the message's row maxima
    (the encoder always does): no LDS, eight points per MFMA tile, the VN activation on accumulator registers.  Same operands, products and
    order as gemm_vn_smallk_kernel / gemm_vn_kernel => IDENTICAL bits.  LS_OPT_GLOB_FUSE = 2 makes the operator export take the exact row maxima
    first (what the tiled kernels compute for themselves), i.e. the streaming kernel; 1 = the tiled kernels."""
    from livingscenes_amd import _lib
    from livingscenes_amd.model_utils import Shape_Prior
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=_dev())
    hip = sp.hip_model()
    g = torch.Generator().manual_seed(11)
    for layer, B, N, scale in ((2, 64, 512, 1.0), (3, 5, 512, 30.0), (2, 3, 64, 1e-3), (3, 1, 8 * 67, 1.0)):
        C = ecfg["feat_dim"][layer]
        msg = (torch.randn(B, N, 3, C, generator=g) * scale * torch.rand(B, N, 1, 1, generator=g)).to(_dev())   # rows of very different magnitude
        a = hip.vn_lna_global(layer, msg)
        prev = hip.set_option(_lib.OPT_GLOB_FUSE, 2)
        b = hip.vn_lna_global(layer, msg)
        hip.set_option(_lib.OPT_GLOB_FUSE, prev)
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (layer, B, N, float((a - b).abs().max()))


def _run_json(cmd, env, timeout=900):
    import json
    import subprocess
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:]
    assert p.stdout.rstrip().endswith(lines[-1]), "the JSON line must be the last thing on stdout"
    return [json.loads(ln) for ln in lines]


def test_bench_gpus_2_runs_two_ranks():
    """`python bench.py --gpus 2` (the driver's command form) must START two ranks -- round 3's bench parsed --gpus and never used it.
    On the one-GPU test box the two ranks share cuda:0 and rendezvous over gloo (LS_BENCH_BACKEND=gloo: a logic check of the N > 1 path,
    not a measurement): n_gpus == 2, one per_rank row per rank, value == 2 ranks x 64 instances x steps / the max-over-ranks time, the
    handles of every rank bit-identical, and the line says it was a dry run."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    (line,) = _run_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-profile",
                         "--cpu-instances", "0", "--no-fma-variant", "--inflight", "2"], env)
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["steps"] == 4
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["objects"] == 64 * 4 for r in line["per_rank"])
    assert line["devices"] == [0, 0] and "dry run" in line["collective_backend"] and line["rccl_ranks"] == 0
    assert abs(line["value"] - 2 * 64 * 4 / (line["ms_per_step"] * 4e-3)) < 1e-6 * line["value"]
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in line["per_rank"]) - 1e-3      # MAX over ranks, never the mean
    assert line["check"]["handles_bit_identical"].startswith("True") and line["check"]["rotations_proper"]
    # each rank pinned to its own cores (livingscenes_amd/launch.py: bind_rank), and the line shows a host-bound rank: max-over-ranks enqueue time
    from livingscenes_amd import launch
    cpu_sets = [set(launch.parse_cpulist(r["cpus"])) for r in line["per_rank"]]
    assert all(cpu_sets), line["per_rank"]
    if len(os.sched_getaffinity(0)) >= 2:
        assert not (cpu_sets[0] & cpu_sets[1]), line["per_rank"]
    assert line["config"]["host_enqueue_basis"] == "max over ranks"
    assert abs(line["config"]["host_enqueue_ms_per_step"] - max(r["host_enqueue_ms_per_step"] for r in line["per_rank"])) < 2e-3
    # one rank, RCCL initialised (the "nccl" backend with a single rank: the only RCCL leg a one-GPU box can run)
    env1 = dict(env, LS_BENCH_FORCE_DIST="1")
    env1.pop("LS_BENCH_BACKEND")
    (one,) = _run_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-profile",
                        "--cpu-instances", "0", "--no-fma-variant", "--inflight", "2"], env1)
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 1 and one["collective_backend"] == "rccl" and len(one["per_rank"]) == 1


def test_configs_sharded_gpus_2_runs_two_ranks():
    """scripts/configs_sharded.py --gpus 2: the same self-launch for configs[3] / [4]; two gloo ranks on the one GPU reproduce the
    single-process counts (registrations, meshes, dense instances)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = ["--scenes", "1", "--dense-instances", "2", "--no-mesh"]
    two = _run_json([sys.executable, os.path.join(root, "scripts", "configs_sharded.py"), "--gpus", "2"] + args, env)
    one = _run_json([sys.executable, os.path.join(root, "scripts", "configs_sharded.py")] + args, env)
    assert [d["n_gpus"] for d in two] == [2, 2] and [d["n_gpus"] for d in one] == [1, 1]
    for a, b in zip(two, one):
        for k in ("scene_pairs", "instance_encodes", "registrations", "meshes", "instances"):
            assert a.get(k) == b.get(k), (k, a, b)


def test_global_conv_is_reproducible_with_streams_in_flight():
    """Run-time side of the packed-fp32 build guard on pointwise.hip (profiles/r6_final/pk_guard_ab.txt): a library whose glob_mean_gemv_kernel
    holds compiler-formed v_pk_fma_f32 is bit-stable alone and WRONG with streams in flight (181 of 240 launches at layer 6, errors to 17 %).  The release
    library must return the same bits from every one of 8 streams x 20 rounds, at the three kinds of global-conv layer (streaming 64-channel kernel, tiled
    kernel with column sums absent, wide rows)."""
    from livingscenes_amd.model_utils import Shape_Prior
    dev = _dev()
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
    m = sp.hip_model()
    g = torch.Generator().manual_seed(3)
    feats = {2: torch.randn(64, 512, 3, 64, generator=g) * 0.1, 4: torch.randn(64, 128, 3, 128, generator=g) * 0.1, 6: torch.randn(64, 32, 3, 512, generator=g) * 0.1}
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    for layer, f in feats.items():
        f = f.to(dev)
        ref = m.vn_lna_global(layer, f).clone()
        torch.cuda.synchronize()
        for _ in range(20):
            outs = []
            for st in streams:
                with torch.cuda.stream(st):
                    outs.append(m.vn_lna_global(layer, f))
            torch.cuda.synchronize()
            assert all(torch.equal(o, ref) for o in outs), f"global conv of layer {layer} differs between streams"
