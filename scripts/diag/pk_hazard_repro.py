"""Reproducer of the reproducibility defect of DESIGN 4.3 (compiler-formed packed fp32 math in the attention kernel).

The attention edge kernel of layer 2 (table path: edge_attn_v4_kernel<16, 1>) runs on STATIC tables from 8 streams at once, each launch preceded on its
stream by two unrelated GEMMs (the matrix-core traffic the defect needs); every output is compared bit for bit with a launch made alone.
With the release library (edge.hip built with -fno-slp-vectorize) the count is 0.  With a library whose edge.hip was built WITH the SLP vectoriser
(scripts/diag/pk_hazard_repro.sh builds it) a few launches of 48 differ.  LS_LIB_PATH selects the library, LS_GEMM_MODE the decoy GEMM's arithmetic.
    python scripts/diag/pk_hazard_repro.py [reps]      -> "<n> of <m> outputs differ" (exit code 0 either way: a measurement, not a test)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import synth, ops, packing, _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
d = torch.device("cuda:0")
cfg = synth.default_encoder_cfg()
desc, blob = packing.pack_model(synth.make_encoder_weights(cfg, 0), cfg, None, None)
m = ops.HipModel(desc, blob, d)
m.set_option(_lib.OPT_EDGE_FUSE_Q, 0)          # the table path: edge_attn_v4_kernel (the fused kernel forms its packed ops on purpose)
B, N = 16, 1024
x = synth.make_instances(B, N, seed=21, rigid=False)
x = (x - x.mean(-1, keepdim=True)) / 1.2
z = m.encode(x.to(d), pre_normalised=True, trace=True)
knn_l, fps_l = z[4], z[5]
f1 = m.edgeconv(1, m.edgeconv(0, x.transpose(1, 2).contiguous().to(d), knn_l[0]), knn_l[1])
args = (2, f1, knn_l[2], fps_l[0])
nbytes = _lib.load().ls_vn_edgeconv_workspace_bytes(m._h, 2, B, f1.shape[1], knn_l[2].shape[1], 1)
NS = 8
streams = [torch.cuda.Stream(device=d) for _ in range(NS)]
wss = [torch.zeros(nbytes, dtype=torch.uint8, device=d) for _ in range(NS + 1)]
ref = m.edgeconv(*args, _ws=wss[NS])            # tables + edge kernel, alone
torch.cuda.synchronize()
for i in range(NS): wss[i].copy_(wss[NS])
m.set_option(_lib.OPT_DEBUG_EDGE, 2)            # from here on: the edge kernel only, on the tables already in the workspace
g = torch.Generator().manual_seed(1)
decoy = ((torch.randn(B * 1024 * 3, 32, generator=g) * 0.3).to(d), (torch.randn(256, 32, generator=g) * 0.1).to(d),
         (torch.randn(B * 512 * 3, 32, generator=g) * 0.3).to(d), (torch.randn(384, 32, generator=g) * 0.1).to(d))
alone = m.edgeconv(*args, _ws=wss[0])
torch.cuda.synchronize()
assert torch.equal(alone, ref), "static-table launch differs from the full operator when run alone"
bad = total = 0
lanes = {}
for rep in range(reps):
    outs = []
    for i, s in enumerate(streams):
        s.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(s):
            ops.gemm(decoy[0], decoy[1]); ops.gemm(decoy[2], decoy[3])
            outs.append(m.edgeconv(*args, _ws=wss[i]))
    torch.cuda.synchronize()
    for o in outs:
        total += 1
        if not torch.equal(o, ref):
            bad += 1
            pts = (o != ref).flatten(2).any(-1).nonzero()           # (instance, point) pairs whose values moved
            for b_, n_ in pts.tolist():
                q = ((b_ * o.shape[1] + n_) % 16) // 4               # 4 points per wave at Co = 64: 16-lane group of the point inside its wave
                lanes[q] = lanes.get(q, 0) + 1
print(f"lib={os.environ.get('LS_LIB_PATH', 'release')} gemm_mode={os.environ.get('LS_GEMM_MODE', 'h2')}: {bad} of {total} outputs differ"
      + (f"; moved points by 16-lane group of their wave {dict(sorted(lanes.items()))}" if lanes else ""))
