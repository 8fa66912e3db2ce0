"""Which per-point scalar of the attention kernel differs?  (debug build of edge.hip with one dump buffer per stream)"""
import os, sys, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["LS_DEBUG_EDGE"] = "notab"
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
args = (2, f1, knn_l[2], fps_l[0])
lib = _lib.load()
lib.ls_debug_set_edge_buf.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
total = B * 512
nbytes = lib.ls_vn_edgeconv_workspace_bytes(m._h, 2, B, 1024, 512, 1)
os.environ.pop("LS_DEBUG_EDGE")
ws0 = torch.zeros(nbytes, dtype=torch.uint8, device=d)
cur = torch.cuda.current_stream(d)
dref = torch.zeros(total, 128, device=d)
lib.ls_debug_set_edge_buf(ctypes.c_void_p(cur.cuda_stream), ctypes.c_void_p(dref.data_ptr()))
ref = m.edgeconv(*args, _ws=ws0)
torch.cuda.synchronize()
os.environ["LS_DEBUG_EDGE"] = "notab"
streams = [torch.cuda.Stream(device=d) for _ in range(8)]
dbgs = [torch.zeros(total, 128, device=d) for _ in range(8)]
wss = [ws0.clone() for _ in range(8)]
for s, b in zip(streams, dbgs):
    lib.ls_debug_set_edge_buf(ctypes.c_void_p(s.cuda_stream), ctypes.c_void_p(b.data_ptr()))
g = torch.Generator().manual_seed(1)
decoy = ((torch.randn(B * 1024 * 3, 32, generator=g) * 0.3).to(d), (torch.randn(256, 32, generator=g) * 0.1).to(d),
         (torch.randn(B * 512 * 3, 32, generator=g) * 0.3).to(d), (torch.randn(384, 32, generator=g) * 0.1).to(d))
names = {0: "ssq_lane", 16: "inv_q", 32: "invk", 48: "raw_score", 64: "ssk", 80: "ex", 96: "score", 112: "mx"}
found = 0
for rep in range(30):
    outs = []
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            ops.gemm(decoy[0], decoy[1]); ops.gemm(decoy[2], decoy[3])
            outs.append(m.edgeconv(*args, _ws=wss[i]))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        if not torch.equal(o, ref):
            pts = (o != ref).reshape(total, -1).any(1).nonzero().flatten().tolist()
            dd = dbgs[i] != dref
            dpts = dd.any(1).nonzero().flatten().tolist()
            print(f"rep {rep} stream {i}: output points {pts[:6]} (n={len(pts)}); dump rows differing {dpts[:6]} (n={len(dpts)})")
            for p in pts[:2]:
                cols = dd[p].nonzero().flatten().tolist()
                print("   point", p, "dump cols", cols[:24])
                for c in cols[:6]:
                    base = max(k for k in names if k <= c)
                    print(f"      {names[base]}[{c - base}]: got {dbgs[i][p, c].item():.9g} ref {dref[p, c].item():.9g}")
            found += 1
    if found >= 5:
        break
print("done, found", found)
