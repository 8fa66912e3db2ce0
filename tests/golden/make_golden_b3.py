#!/usr/bin/env python3
"""Fixture that PINS the B = 3 decision with the reference's own code.

    python tests/golden/make_golden_b3.py          # writes tests/golden/encoder_b3.npz

The reference's get_graph_feature calls torch.cross WITHOUT a dim (/root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:157):
the product is taken over the first axis of size 3, which is the xyz axis (dim 2 of [B, C, 3, N, K]) for every batch size EXCEPT B = 3, where
it is the batch axis -- a batched encode of three instances then mixes the instances.  This build always crosses over xyz.  The fixture holds the
reference's VecDGCNN_att.forward (small configuration, the build's synthetic weights, the pytorch3d shims of make_golden.py) on three instances
  * ONE AT A TIME (B = 1, three calls): what three independent instances encode to -- the behaviour the build reproduces at B = 3, and
  * as ONE batch of B = 3: the reference's own batched result, which differs (kept so that the divergence is a recorded fact, not a claim).
Data only: inputs and outputs, no reference source text."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (the loader / shim helpers; nothing of the other fixtures is rewritten)
from livingscenes_amd import synth  # noqa: E402


def main():
    torch.set_num_threads(8)
    mg.install_stubs()
    mg.load_by_path("vec_layers", "lib_shape_prior/core/lib/vec_sim3/vec_layers.py")
    att = mg.load_by_path("ref_vec_dgcnn_atten", "lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py")
    cfg = synth.small_encoder_cfg()
    net = att.VecDGCNN_att(**cfg).eval()
    net.load_state_dict(synth.make_encoder_weights(cfg, seed=7), strict=True)
    x = synth.make_instances(3, 128, seed=13, rigid=False)
    x = x - x.mean(-1, keepdim=True)
    out = {"x": x}
    names = ("center", "scale", "z_so3", "z_inv")
    with torch.no_grad():
        singles = []
        for b in range(3):
            mg.CAPTURE["knn"].clear(), mg.CAPTURE["fps"].clear()
            singles.append(net(x[b:b + 1]))
            for i, t in enumerate(mg.CAPTURE["knn"]):
                out[f"single{b}_knn_idx_{i}"] = t.to(torch.int32)
            for i, t in enumerate(mg.CAPTURE["fps"]):
                out[f"single{b}_fps_idx_{i}"] = t.to(torch.int32)
        for k, name in enumerate(names):
            out["single_" + name] = torch.cat([s[k] for s in singles], 0)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # torch.cross without dim: the deprecation warning IS the quirk
            batched = net(x)
        for k, name in enumerate(names):
            out["batched_" + name] = batched[k]
    np.savez_compressed(os.path.join(HERE, "encoder_b3.npz"), **mg.t2n(out))
    for name in names:
        a, b = out["single_" + name], out["batched_" + name]
        print(name, "per-instance vs batched B=3: rel", float((a - b).abs().max() / a.abs().max()))


if __name__ == "__main__":
    main()
