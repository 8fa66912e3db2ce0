"""Does the ORDER in which destination points are processed matter to the attention gather kernel (L2 / HBM over-fetch)?
Permute the destination points of layers 2..4 (knn rows + FPS rows together) into Morton order of their xyz and time the gather."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from livingscenes_amd import _lib, ops, packing, synth
d = torch.device("cuda:0")
cfg = synth.default_encoder_cfg()
w = synth.make_encoder_weights(cfg, 0)
desc, blob = packing.pack_model(w, cfg, None, None)
m = ops.HipModel(desc, blob, d)
B, N = 64, 1024
scene = synth.make_scene_pair(B // 2, N, seed=1000)
x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(d)
pts, cen, sc0 = ops.encode_prologue(x)
z = m.encode(x, trace=True)
knn_l, fps_l = z[4], z[5]
L, g0, ds = cfg["num_layers"], cfg["res_global_start_layer"], cfg["down_sample_layers"]
src, rows, xyz = [None] * L, [None] * L, [None] * L
cur, level, cur_pts = pts, 0, pts
for i in range(L):
    if i in ds:
        rows[i] = fps_l[level]; level += 1
        cur_pts = torch.gather(cur_pts, 1, rows[i].long()[..., None].expand(-1, -1, 3)).contiguous()
    xyz[i] = cur_pts
    src[i] = cur
    msg = m.edgeconv(i, cur, knn_l[i], rows[i])
    cur = m.vn_lna_global(i, msg) if i >= g0 else msg

def morton(p):   # [B,n,3] -> permutation [B,n] by 10-bit-per-axis Morton code
    q = ((p - p.min(1, keepdim=True)[0]) / (p.max(1, keepdim=True)[0] - p.min(1, keepdim=True)[0] + 1e-9) * 1023).long()
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[..., 0]) | (spread(q[..., 1]) << 1) | (spread(q[..., 2]) << 2)
    return code.argsort(1)

lib = _lib.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for i in (1, 2, 3, 4, 5):
    Nd = knn_l[i].shape[1]
    nb = lib.ls_vn_edgeconv_workspace_bytes(m._h, i, B, src[i].shape[1], Nd, int(rows[i] is not None))
    ws = torch.empty(nb, dtype=torch.uint8, device=d)
    os.environ["LS_DEBUG_EDGE"] = "tabonly"; m.edgeconv(i, src[i], knn_l[i], rows[i], _ws=ws)
    os.environ["LS_DEBUG_EDGE"] = "notab"
    perm = morton(xyz[i])
    ident = torch.arange(Nd, device=d)[None].expand(B, -1)
    r_id = rows[i] if rows[i] is not None else None
    # with dst_rows == None the kernel takes destination n = source n: emulate a permuted order by passing rows = perm
    knn_p = torch.gather(knn_l[i], 1, perm[..., None].expand(-1, -1, 16)).contiguous()
    rows_p = (torch.gather(rows[i], 1, perm) if rows[i] is not None else perm.int()).contiguous()
    t0 = timeit(lambda: m.edgeconv(i, src[i], knn_l[i], r_id, _ws=ws))
    if rows[i] is not None:
        t1 = timeit(lambda: m.edgeconv(i, src[i], knn_p, rows_p, _ws=ws))
        print(f"layer {i}: stored order {t0:.1f} us, Morton order {t1:.1f} us")
    else:
        print(f"layer {i}: stored order {t0:.1f} us (no dst_rows: order fixed by the table layout)")
    os.environ.pop("LS_DEBUG_EDGE")
