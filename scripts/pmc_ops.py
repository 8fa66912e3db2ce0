#!/usr/bin/env python3
"""Run every encoder operator of the bench workload (B = 64 instances x 1024 points, released widths) ALONE, a few times, in a
known order, so that rocprofv3 --pmc counters can be attributed to (operator, layer) -- the same kernel name and grid occur in
several layers, so (name, grid) keys cannot.  Each operator call is preceded by a MARKER (a tiny at::native elementwise kernel;
the library itself never launches one), and the ordered list of labels is written as a manifest; scripts/pmc_ops_summary.py
splits the dispatch stream of the counter CSVs at the markers.

    for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do   # separate passes (MI355X_MICROARCH.md)
        rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc/<tag> -o p --output-format csv -- python scripts/pmc_ops.py --manifest gpurun_out/pmc/manifest.json
    done
    python scripts/pmc_ops_summary.py gpurun_out/pmc > profiles/pmc_latest.json

The operators are the C-ABI per-layer exports (ls_knn_f32, ls_vn_edgeconv_* split into its table
GEMM and its gather kernel by LS_OPT_DEBUG_EDGE, ls_vn_lna_f32, ls_encoder_tail_f32, FPS, prologue) on the REAL per-layer tensors of
the bench batch (computed once by the same operators).  Without rocprofv3 the script also prints hipEvent timings per operator."""
import argparse
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from livingscenes_amd import _lib, ops, packing, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--manifest", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
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
    lib = _lib.load()

    pts, cen, sc0 = ops.encode_prologue(x)
    z = m.encode(x, trace=True)
    knn_l, fps_l = z[4], z[5]
    # real per-layer tensors
    src, msg, out, rows = [None] * L, [None] * L, [None] * L, [None] * L
    cur, level, cur_pts, pts_l = pts, 0, pts, [pts]
    for i in range(L):
        if i in ds:
            rows[i] = fps_l[level]
            level += 1
            cur_pts = torch.gather(cur_pts, 1, rows[i].long()[..., None].expand(-1, -1, 3)).contiguous()
            pts_l.append(cur_pts)
        src[i] = cur
        msg[i] = m.edgeconv(i, cur, knn_l[i], rows[i])
        out[i] = m.vn_lna_global(i, msg[i]) if i >= g0 else msg[i]
        cur = out[i]
    torch.cuda.synchronize()

    # the operator list is built FIRST (hints gathered etc.), so that no at::native kernel other than the markers runs between operators
    todo = [("prologue[0]", lambda: ops.encode_prologue(x))]
    for lv in range(len(ds)):
        todo.append((f"fps[{lv}]", lambda lv=lv: ops.fps(pts_l[lv], pts_l[lv + 1].shape[1])))
    keep = []
    for i in range(L):
        f = src[i].reshape(B, -1, 3, 1) if i == 0 else src[i]
        Cin = 1 if i == 0 else cfg["feat_dim"][i - 1]
        todo.append((f"knn[{i}]", lambda i=i, f=f: ops.knn(f, f, 16, dst_rows=rows[i])))
        if i == 0:
            todo.append(("edge_l0[0]", lambda: m.edgeconv(0, src[0], knn_l[0])))
        else:
            nb = lib.ls_vn_edgeconv_workspace_bytes(m._h, i, B, src[i].shape[1], knn_l[i].shape[1], int(rows[i] is not None))
            ws = torch.empty(nb, dtype=torch.uint8, device=d)
            keep.append(ws)
            kind = "edge_attn" if i >= cfg["atten_start_layer"] else "edge_pool"

            def tab(i=i, ws=ws):
                m.set_option(_lib.OPT_DEBUG_EDGE, 1)      # the table GEMM only
                m.edgeconv(i, src[i], knn_l[i], rows[i], _ws=ws)
                m.set_option(_lib.OPT_DEBUG_EDGE, 0)

            def gather(i=i, ws=ws):
                m.set_option(_lib.OPT_DEBUG_EDGE, 2)      # the edge kernel only, on the tables already in the workspace
                m.edgeconv(i, src[i], knn_l[i], rows[i], _ws=ws)
                m.set_option(_lib.OPT_DEBUG_EDGE, 0)
            todo += [(f"gemm_edge[{i}]", tab), (f"{kind}[{i}]", gather)]
        if i >= g0:
            todo.append((f"global_conv[{i}]", lambda i=i: m.vn_lna_global(i, msg[i])))
    todo.append(("tail[0]", lambda: m.encoder_tail(out[L - 1], cen, sc0)))
    torch.cuda.synchronize()

    manifest, timings = [], {}
    mark = torch.zeros(64, device=d)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for label, fn in todo:
        for r in range(args.reps):
            mark.add_(1.0)                                  # MARKER kernel (at::native::...)
            manifest.append(label)
            if r == args.reps - 1:
                ev[0].record()
            fn()
            if r == args.reps - 1:
                ev[1].record()
        torch.cuda.synchronize()
        timings[label] = ev[0].elapsed_time(ev[1]) * 1e3
    if args.manifest:
        os.makedirs(os.path.dirname(os.path.abspath(args.manifest)), exist_ok=True)
        with open(args.manifest, "w") as fh:
            json.dump({"ops": manifest, "reps": args.reps, "batch": B, "points": N, "hipevent_us_last_rep": timings,
                       "under_profiler": bool(os.environ.get("ROCPROFILER_REGISTER_ROOT") or os.environ.get("ROCP_TOOL_LIBRARIES"))}, fh, indent=1)
    print(json.dumps({k: round(v, 1) for k, v in timings.items()}))


if __name__ == "__main__":
    main()
