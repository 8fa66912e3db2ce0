"""Dump the attention layers' outputs (ls_vn_edgeconv_attn_f32 on the HIP path's own layer inputs) to an npz: run once per build /
environment (LS_EDGE_FUSE_Q=0|1) and diff the files -- quantifies a bit-identity failure between the fused-destination kernel and the
table path.    python scripts/diag/attn_paths_diff.py out.npz ; python scripts/diag/attn_paths_diff.py a.npz b.npz (compare)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    for k in a.files:
        x, y = a[k], b[k]
        d = np.abs(x.astype(np.float64) - y)
        print(k, "identical" if np.array_equal(x, y) else f"differs: {int((x != y).sum())} of {x.size} values, max |d| {d.max():.3e} (max |x| {np.abs(x).max():.3e}), rows {np.unique(np.nonzero(x != y)[0])[:8]}")
    sys.exit(0)
from livingscenes_amd import ops, packing, synth
dev = torch.device("cuda:0")
cfg = synth.default_encoder_cfg()
w = synth.make_encoder_weights(cfg, 0)
desc, blob = packing.pack_model(w, cfg, None, None)
m = ops.HipModel(desc, blob, dev)
B, N = 8, 1024
x = synth.make_instances(B, N, seed=21, rigid=False)
x = (x - x.mean(-1, keepdim=True)) / 1.2
z = m.encode(x.to(dev), pre_normalised=True, trace=True)
knn_l, fps_l = z[4], z[5]
f = m.edgeconv(1, m.edgeconv(0, x.transpose(1, 2).contiguous().to(dev), knn_l[0]), knn_l[1])
out, lvl = {}, 0
for i in range(2, cfg["num_layers"]):
    rows = fps_l[lvl] if i in cfg["down_sample_layers"] else None
    lvl += i in cfg["down_sample_layers"]
    msg = m.edgeconv(i, f, knn_l[i], rows)
    out[f"msg{i}"] = msg.cpu().numpy()
    f = m.vn_lna_global(i, msg)
np.savez(sys.argv[1], **out)
