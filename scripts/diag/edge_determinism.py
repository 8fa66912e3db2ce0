"""Race hunting: the attention edge-conv operator (table GEMM + gather kernel) from 8 streams at once, in variants."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import synth, ops, packing, _lib
d = torch.device("cuda:0")
cfg = synth.default_encoder_cfg()
w = synth.make_encoder_weights(cfg, 0)
desc, blob = packing.pack_model(w, cfg, None, None)
m = ops.HipModel(desc, blob, d)
B, N = 16, 1024
x = synth.make_instances(B, N, seed=21, rigid=False)
x = (x - x.mean(-1, keepdim=True)) / 1.2
z = m.encode(x.to(d), pre_normalised=True, trace=True)
knn_l, fps_l = z[4], z[5]
f1 = m.edgeconv(1, m.edgeconv(0, x.transpose(1, 2).contiguous().to(d), knn_l[0]), knn_l[1])
layer = int(os.environ.get("LAYER", "2"))
mode = os.environ.get("MODE", "full")
if layer == 2:
    args = (2, f1, knn_l[2], fps_l[0])
elif layer == 1:
    f0 = m.edgeconv(0, x.transpose(1, 2).contiguous().to(d), knn_l[0])
    args = (1, f0, knn_l[1], None)
else:
    raise SystemExit("LAYER 1|2")
nbytes = _lib.load().ls_vn_edgeconv_workspace_bytes(m._h, args[0], B, args[1].shape[1], args[2].shape[1], int(args[3] is not None))
NS = int(os.environ.get('NSTREAMS', '8'))
streams = [torch.cuda.Stream(device=d) for _ in range(NS)]
wss = [torch.zeros(nbytes, dtype=torch.uint8, device=d) for _ in range(9)]
ref = m.edgeconv(*args, _ws=wss[8])
torch.cuda.synchronize()
ref_ws = wss[8].clone()
decoy = None
if mode == "decoy":   # attention on STATIC tables, but an unrelated table-shaped GEMM runs right before it on the same stream
    mode = "notab"
    g = torch.Generator().manual_seed(1)
    decoy = ((torch.randn(B * 1024 * 3, 32, generator=g) * 0.3).to(d), (torch.randn(256, 32, generator=g) * 0.1).to(d),
             (torch.randn(B * 512 * 3, 32, generator=g) * 0.3).to(d), (torch.randn(384, 32, generator=g) * 0.1).to(d))
if mode == "notab":
    for i in range(8): wss[i].copy_(wss[8])
    os.environ["LS_DEBUG_EDGE"] = "notab"
if mode == "tabonly":
    os.environ["LS_DEBUG_EDGE"] = "tabonly"
torch.cuda.synchronize()
bad = 0
for rep in range(6 * (8 // NS)):
    outs = []
    for i, s in enumerate(streams):
        s.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(s):
            if decoy is not None:
                ops.gemm(decoy[0], decoy[1]); ops.gemm(decoy[2], decoy[3])
            outs.append(m.edgeconv(*args, _ws=wss[i]))
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        if mode == "tabonly":
            if not torch.equal(wss[k], ref_ws):
                bad += 1
        elif not torch.equal(o, ref):
            bad += 1
            tab_same = torch.equal(wss[k], ref_ws)
            print(f"  rep {rep} stream {k}: {int((o != ref).sum())} floats differ; tables identical: {tab_same}")
print(f"layer {layer} mode {mode} BF16X3={os.environ.get('LS_GEMM_BF16X3','1')}: {bad} of 48 outputs differ")
