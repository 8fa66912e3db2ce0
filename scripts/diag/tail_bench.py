import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from livingscenes_amd import ops, packing, synth
dev = torch.device("cuda:0")
cfg = synth.default_encoder_cfg()
w = synth.make_encoder_weights(cfg, 0)
desc, blob = packing.pack_model(w, cfg, None, None)
m = ops.HipModel(desc, blob, dev)
B = 64
f = torch.randn(B, 32, 3, cfg["feat_dim"][-1], device=dev)
for _ in range(5): m.encoder_tail(f)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): m.encoder_tail(f)
e1.record(); torch.cuda.synchronize()
print("encoder_tail op (GEMM + tail kernel):", e0.elapsed_time(e1) / 50 * 1e3, "us")
