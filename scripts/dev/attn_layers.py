#!/usr/bin/env python3
"""dev: the attention layers 2 - 4 of the bench batch (B = 64 x 1024 points, released widths) ALONE -- table GEMM once, then the attention kernel `reps`
times (LS_OPT_DEBUG_EDGE = 2) -- under each LS_OPT_EDGE_STAGED mode given: hipEvent time per launch.  Run it under rocprofv3 --pmc for counters
(scripts/dev/attn_counters.sh)."""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import _lib, ops, packing, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="0,2")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--layers", default="2,3,4")
    args = ap.parse_args()
    d = torch.device("cuda:0")
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 0)
    desc, blob = packing.pack_model(w, cfg, None, None)
    m = ops.HipModel(desc, blob, d)
    B, N = args.batch, 1024
    scene = synth.make_scene_pair(B // 2, N, seed=1000)
    x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(d)
    L, g0, ds = cfg["num_layers"], cfg["res_global_start_layer"], cfg["down_sample_layers"]
    pts, cen, sc0 = ops.encode_prologue(x)
    z = m.encode(x, trace=True)
    knn_l, fps_l = z[4], z[5]
    src, rows = [None] * L, [None] * L
    cur, level = pts, 0
    for i in range(L):
        if i in ds:
            rows[i] = fps_l[level]
            level += 1
        src[i] = cur
        msg = m.edgeconv(i, cur, knn_l[i], rows[i])
        cur = m.vn_lna_global(i, msg) if i >= g0 else msg
    torch.cuda.synchronize()
    res = {}
    for mode in [int(v) for v in args.modes.split(",")]:
        m.set_option(_lib.OPT_EDGE_STAGED, mode)
        for i in [int(v) for v in args.layers.split(",")]:
            m.set_option(_lib.OPT_DEBUG_EDGE, 0)
            ref = m.edgeconv(i, src[i], knn_l[i], rows[i])          # table + attention (also warms up)
            m.set_option(_lib.OPT_DEBUG_EDGE, 2)                    # the attention kernel only, on the tables in the workspace
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
            for a, b in evs:
                a.record()
                out = m.edgeconv(i, src[i], knn_l[i], rows[i])
                b.record()
            torch.cuda.synchronize()
            m.set_option(_lib.OPT_DEBUG_EDGE, 0)
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
            res[(mode, i)] = ts[len(ts) // 2]
            assert os.environ.get("LS_LIB_PATH") or torch.equal(out, ref)      # (timing-variant libraries compute garbage on purpose)
            print(f"mode {mode} layer {i}: attention alone {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})", flush=True)


if __name__ == "__main__":
    main()
