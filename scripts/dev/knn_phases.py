#!/usr/bin/env python3
"""dev: the k-NN builds of layers 1 - 4 of the bench batch ALONE, for timing-variant libraries of knn_mfma.hip (LS_KF_STOP = 1 / 2 / 3: the fused kernel
up to the threshold / the pair list / the exact distances -- wrong results by design, so the layer inputs come from a file written with the release library).
    python scripts/dev/knn_phases.py --dump /tmp/knn_in.pt                      (release library)
    LS_LIB_PATH=... python scripts/dev/knn_phases.py --time /tmp/knn_in.pt       -> us per call and layer (image + fused kernel), hipEvent median"""
import argparse
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import ops, packing, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump")
    ap.add_argument("--time")
    ap.add_argument("--reps", type=int, default=9)
    args = ap.parse_args()
    d = torch.device("cuda:0")
    if args.dump:
        cfg = synth.default_encoder_cfg()
        desc, blob = packing.pack_model(synth.make_encoder_weights(cfg, 0), cfg, None, None)
        m = ops.HipModel(desc, blob, d)
        B, N = 64, 1024
        scene = synth.make_scene_pair(B // 2, N, seed=1000)
        x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(d)
        L, g0, ds = cfg["num_layers"], cfg["res_global_start_layer"], cfg["down_sample_layers"]
        pts, cen, sc0 = ops.encode_prologue(x)
        z = m.encode(x, trace=True)
        knn_l, fps_l = z[4], z[5]
        cur, level, out = pts, 0, {}
        for i in range(5):
            rows = None
            if i in ds:
                rows = fps_l[level]
                level += 1
            if i >= 1:
                out[i] = (cur.cpu(), None if rows is None else rows.cpu(), knn_l[i].cpu())
            msg = m.edgeconv(i, cur, knn_l[i], rows)
            cur = m.vn_lna_global(i, msg) if i >= g0 else msg
        torch.save(out, args.dump)
        return
    data = torch.load(args.time)
    for i, (f, rows, ref) in sorted(data.items()):
        f = f.to(d)
        rows = None if rows is None else rows.to(d)
        idx = ops.knn(f, f, 16, dst_rows=rows)
        ok = bool(torch.equal(idx.cpu(), ref))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in evs:
            a.record()
            ops.knn(f, f, 16, dst_rows=rows)
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        print(f"layer {i}: {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})  lists == release: {ok}", flush=True)


if __name__ == "__main__":
    main()
