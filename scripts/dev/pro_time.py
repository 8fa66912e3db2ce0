#!/usr/bin/env python3
"""dev: ls_encode_prologue_f32 alone on the bench batch, 30 times (under rocprofv3 --kernel-trace --stats for the kernel duration; LS_PRO_STOP timing variants)."""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import ops, synth
d = torch.device("cuda:0")
scene = synth.make_scene_pair(32, 1024, seed=1000)
x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(d)
for _ in range(30):
    pts, cen, sc0 = ops.encode_prologue(x)
torch.cuda.synchronize()
print("scale0[:4]", sc0[:4].tolist())
