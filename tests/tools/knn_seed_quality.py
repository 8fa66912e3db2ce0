"""How good are the k-NN hints the encoder passes from layer to layer?  (dev tool; CPU, uses the oracle's layer trace)

For every layer whose k-NN is seeded it reports, per query, how many candidates have a canonical distance <= the K-th
distance among the hints (= what the MFMA sweep must hand to the finish kernel).  Usage: python tests/tools/knn_seed_quality.py [B]
"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from livingscenes_amd import synth
from oracle import net

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = synth.default_encoder_cfg()
w = net.as_params(synth.make_encoder_weights(cfg, seed=0))
x = synth.make_instances(B, 1024, seed=0)
x = (x[0] if isinstance(x, (tuple, list)) else x).float()
tr = {}
net.shape_prior_encode(w, cfg, x, tr)
L = cfg["num_layers"]


def rows(f):  # [B,C,3,N] -> [B,N,3C] in canonical order j = c*3+x
    return f.permute(0, 3, 1, 2).reshape(f.shape[0], f.shape[3], -1)


def survivors(src, dst, hints):
    d = torch.cdist(dst.double(), src.double()) ** 2                      # [B,Nd,Ns]
    ok = hints >= 0
    hd = torch.gather(d, 2, hints.clamp(min=0).long())
    hd[~ok] = float("inf")
    # distinct hints only
    kth = []
    for b in range(d.shape[0]):
        for q in range(d.shape[1]):
            u = torch.unique(hints[b, q][ok[b, q]])
            v = torch.sort(d[b, q, u.long()])[0]
            kth.append(v[15] if len(v) >= 16 else torch.tensor(float("inf"), dtype=torch.float64))
    kth = torch.stack(kth).reshape(d.shape[0], d.shape[1], 1)
    exact = (d <= kth).sum(-1).flatten().numpy()
    # bf16 filter on centred features: d^ = |q|^2 + |s|^2 - 2 q~.s~, dropped iff d^ - eps (|q|^2+|s|^2) > kth, eps = 2.05 * 2^-8
    mu = src.mean(1, keepdim=True)
    sc, dc = (src - mu).float(), (dst - mu).float()
    S = torch.matmul(dc.bfloat16().float(), sc.bfloat16().float().transpose(1, 2)).double()
    nn = (dc.double() ** 2).sum(-1, keepdim=True) + (sc.double() ** 2).sum(-1).unsqueeze(1)
    dh = nn - 2 * S
    res = {"exact": exact}
    for name, eps in (("fp32", 6 * (src.shape[2] + 4) * 2.0 ** -24), ("bf16", 2.05 * 2.0 ** -8), ("bf16x2", 4.0 * 2.0 ** -16)):
        dd = dh if name == "bf16" else (nn - 2 * torch.matmul(dc.double(), sc.double().transpose(1, 2)))
        res[name] = ((dd - eps * nn) <= kth).sum(-1).flatten().numpy()
    res["d_over_nn"] = float((kth.squeeze(-1)[torch.isfinite(kth.squeeze(-1))] / nn.mean(-1)[torch.isfinite(kth.squeeze(-1))]).median())
    return res


prev = None
for i in range(1, L):
    src, dst, idx = rows(tr[f"src_f_{i}"]), rows(tr[f"dst_f_in_{i}"]), tr[f"knn_idx_{i}"]
    Ns, Nd = src.shape[1], dst.shape[1]
    fidx = tr.get(f"fps_idx_{i}")
    pk, pf = tr[f"knn_idx_{i-1}"], tr.get(f"fps_idx_{i-1}")
    if pf is None:     # previous layer kept its points: its list of the query's source row
        hints = pk if fidx is None else torch.gather(pk, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 16))
        kind = "direct"
    else:              # previous layer down-sampled: 1-hop mapped + 2-hop (same rule as knn_compose_hints_kernel)
        Np = tr[f"src_f_{i-1}"].shape[3]
        hints = -torch.ones(B, Ns, 16, dtype=torch.long)
        for b in range(B):
            inv = -np.ones(Np, np.int64); inv[pf[b].numpy()] = np.arange(len(pf[b]))
            pkb = pk[b].numpy()
            for q in range(Ns):
                h = [v for v in inv[pkb[q]] if v >= 0]
                t = 0
                cap = 32 if os.environ.get("HINTS32") else 16
                while t < len(h) and len(h) < cap:      # breadth first
                    for v in inv[pkb[h[t]]]:
                        if v >= 0 and v not in h and len(h) < cap: h.append(v)
                    t += 1
                if cap == 32:                            # keep the 16 best of the 32 by exact distance
                    srcb, dstb = rows(tr[f"src_f_{i}"])[b], rows(tr[f"dst_f_in_{i}"])[b]
                    dd = ((srcb[h].double() - dstb[q].double()) ** 2).sum(-1)
                    h = [h[j] for j in torch.argsort(dd)[:16].tolist()]
                e = 0
                while len(h) < 16:                      # isolated point: arbitrary distinct rows
                    v = (q + 1 + e * 37) % Ns; e += 1
                    if v not in h: h.append(v)
                hints[b, q, :len(h)] = torch.tensor(h[:16])
        if fidx is not None:
            hints = torch.gather(hints, 1, fidx.long().unsqueeze(-1).expand(-1, -1, 16))
        kind = "composed"
    res = survivors(src, dst, hints)
    print(f"layer {i}: Nd={Nd} Ns={Ns} D={src.shape[2]} hints={kind}  kth/(|q|^2+|s|^2) median {res.pop('d_over_nn'):.3f}")
    for name, s in res.items():
        print(f"    {name:7s} survivors/query mean {s.mean():6.1f} median {np.median(s):4.0f} p90 {np.percentile(s, 90):4.0f} max {s.max():4d}  (of {Ns})")
