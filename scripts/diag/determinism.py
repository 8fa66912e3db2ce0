"""Diagnostic: are the encoder's outputs bit-reproducible (a) run to run on one handle, (b) across handles in flight?"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("LS_GEMM_OVERLAP", "0")
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import synth
from livingscenes_amd.model_utils import Shape_Prior
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
nfl = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sps = [Shape_Prior.from_state(ecfg, dcfg, ew, dw, device=dev) for _ in range(nfl)]
streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
scene = synth.make_scene_pair(32, 1024, seed=1000)
x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)

def cmp(a, b, tag):
    names = ["z_so3", "z_inv", "s", "t"] + [f"knn{i}" for i in range(7)] + [f"fps{i}" for i in range(3)]
    flat_a = list(a[:4]) + list(a[4]) + list(a[5]); flat_b = list(b[:4]) + list(b[4]) + list(b[5])
    bad = []
    for n, u, v in zip(names, flat_a, flat_b):
        if not torch.equal(u, v):
            d = (u.float() - v.float()).abs()
            bad.append((n, int((u != v).sum()), float(d.max())))
    print(tag, "IDENTICAL" if not bad else bad)

with torch.no_grad():
    h = sps[0].hip_model()
    r0 = h.encode(x, trace=True); torch.cuda.synchronize()
    r1 = h.encode(x, trace=True); torch.cuda.synchronize()
    cmp(r0, r1, "same handle, serial:")
    outs = [None] * nfl
    for rep in range(3):
        for i in range(nfl):
            with torch.cuda.stream(streams[i]):
                outs[i] = sps[i].hip_model().encode(x, trace=True)
        torch.cuda.synchronize()
        for i in range(nfl):
            cmp(r0, outs[i], f"rep {rep} handle {i} in flight vs serial:")
