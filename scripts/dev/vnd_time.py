#!/usr/bin/env python3
"""dev: the residual global conv of layer 2 (B = 64 x 512 points x 64 channels) alone, 30 times, through the operator export with LS_OPT_GLOB_FUSE = 2
(row maxima + mean + the streaming kernel).  Run under rocprofv3 --kernel-trace --stats for the kernel's own duration (scripts/dev/vnd_variants.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import _lib, ops, packing, synth
d = torch.device("cuda:0")
cfg = synth.default_encoder_cfg()
desc, blob = packing.pack_model(synth.make_encoder_weights(cfg, 0), cfg, None, None)
m = ops.HipModel(desc, blob, d)
m.set_option(_lib.OPT_GLOB_FUSE, 2)
g = torch.Generator().manual_seed(3)
layer = int(os.environ.get("LAYER", "2"))
npts = {2: 512, 3: 512, 4: 128, 5: 32, 6: 32}[layer]
msg = torch.randn(64, npts, 3, cfg["feat_dim"][layer], generator=g).to(d)
for _ in range(30):
    out = m.vn_lna_global(layer, msg)
torch.cuda.synchronize()
print("ok", float(out.abs().max()))
