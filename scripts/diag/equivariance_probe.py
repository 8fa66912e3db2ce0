import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from livingscenes_amd import synth, ops, packing
dev = torch.device("cuda:0")
def relerr(a, b):
    a = a.detach().cpu().double().numpy(); b = b.detach().cpu().double().numpy()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
cfg = synth.default_encoder_cfg()
w = synth.make_encoder_weights(cfg, 0)
desc, blob = packing.pack_model(w, cfg, None, None)
m = ops.HipModel(desc, blob, dev)
for seed in (21, 22, 23, 24, 25, 26):
    x = synth.make_instances(2, 1024, seed=seed, rigid=False)
    x = x - x.mean(-1, keepdim=True)
    rng = np.random.default_rng(seed)
    R = torch.from_numpy(np.stack([synth._rand_rot(rng) for _ in range(2)]).astype(np.float32))
    sc = torch.tensor([0.7, 1.4])
    xa = torch.einsum("bij,bjn->bin", R, x * sc[:, None, None])
    z0, i0, s0, _, k0, f0 = m.encode(x.to(dev), pre_normalised=True, trace=True)
    z1, i1, s1, _, k1, f1 = m.encode(xa.to(dev), pre_normalised=True, trace=True)
    zr = torch.einsum("bij,bcj->bci", R.to(dev), z0)
    same = [bool(torch.equal(a, b)) for a, b in zip(k0, k1)] + [bool(torch.equal(a, b)) for a, b in zip(f0, f1)]
    frac = [float((a == b).float().mean()) for a, b in zip(k0, k1)]
    print(seed, "z_so3 %.2e z_inv %.2e s %.2e" % (relerr(z1, zr), relerr(i1, i0), relerr(s1, s0 * sc.to(dev))), "graphs equal:", same, "knn agreement", [round(v, 4) for v in frac])
