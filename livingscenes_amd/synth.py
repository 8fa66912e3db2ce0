"""Deterministic synthetic weights and inputs (the released checkpoint and all datasets are absent:
/root/reference/.MISSING_LARGE_BLOBS:1).  Counter-based (numpy Philox keyed by seed and tensor
name), so the same tensors are regenerated bit-for-bit on the GPU box, in tests and in bench.py.

Shapes/hyper-parameters are those of /root/reference/weights/files_backup/model_config.yaml:105-178
(identical to lib_shape_prior/configs/3rscan/dgcnn_attn_inner.yaml:29-44).
"""
import zlib

import numpy as np
import torch


def default_encoder_cfg():
    return dict(
        atten_multi_head_c=16, atten_start_layer=2, c_dim=256, center_pred=True, center_pred_scale=True,
        down_sample_factor=[2, 4, 4], down_sample_layers=[2, 4, 5],
        feat_dim=[32, 32, 64, 64, 128, 256, 512], leak_neg_slope=0.2, num_knn=16, num_layers=7,
        res_global_start_layer=2, scale_factor=64000.0, use_dg=True, use_res_global_conv=True)


def default_decoder_cfg():
    return dict(
        dims=[768] * 8, dropout=list(range(8)), dropout_prob=0.2, latent_dropout=False, latent_in=[4],
        latent_size=256, norm_layers=list(range(8)), pe_dim=257, use_tanh=False, weight_norm=True)


def small_encoder_cfg():
    """Reduced-width/-depth config for fast CPU tests (same structure: pool layers, attention layers,
    one down-sample, residual global conv, all heads)."""
    return dict(
        atten_multi_head_c=16, atten_start_layer=2, c_dim=32, center_pred=True, center_pred_scale=True,
        down_sample_factor=[2], down_sample_layers=[2],
        feat_dim=[32, 32, 32, 64], leak_neg_slope=0.2, num_knn=16, num_layers=4,
        res_global_start_layer=2, scale_factor=64000.0, use_dg=True, use_res_global_conv=True)


def small_decoder_cfg():
    return dict(
        dims=[128] * 4, dropout=None, dropout_prob=0.0, latent_dropout=False, latent_in=[2],
        latent_size=32, norm_layers=list(range(4)), pe_dim=33, use_tanh=False, weight_norm=True)


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _uniform(seed, name, shape, bound):
    r = _rng(seed, name).random(size=shape, dtype=np.float64)
    return torch.from_numpy(((2.0 * r - 1.0) * bound).astype(np.float32))


def encoder_param_shapes(cfg):
    """Ordered {state_dict name: shape} of VecDGCNN_att (vec_dgcnn_atten.py:76-122)."""
    fd, L = cfg["feat_dim"], cfg["num_layers"]
    a0, g0, c = cfg["atten_start_layer"], cfg["res_global_start_layer"], cfg["c_dim"]
    shp = {}
    for i in range(L):
        cin = 3 if i == 0 else 2 * fd[i - 1]
        shp[f"V_list.{i}.lin.weight"] = (fd[i], cin)
        shp[f"V_list.{i}.act.lin_dir.weight"] = (fd[i], fd[i])
    for i in range(L):
        if cfg.get("use_res_global_conv", True) and i >= g0:
            j = i - g0
            shp[f"global_conv_list.{j}.lin.weight"] = (fd[i], 2 * fd[i])
            shp[f"global_conv_list.{j}.act.lin_dir.weight"] = (fd[i], fd[i])
    for i in range(L):
        if i >= a0:
            shp[f"Q_list.{i}.lin.weight"] = (fd[i], fd[i - 1])
            shp[f"Q_list.{i}.act.lin_dir.weight"] = (fd[i], fd[i])
            shp[f"K_list.{i}.lin.weight"] = (fd[i], 2 * fd[i - 1])
            shp[f"K_list.{i}.act.lin_dir.weight"] = (fd[i], fd[i])
    shp["conv_c.lin.weight"] = (c, fd[-1])
    shp["conv_c.act.lin_dir.weight"] = (1, c)
    shp["fc_inv.weight"] = (c, c)
    if cfg.get("center_pred", False):
        h = c // 2
        shp["fc_center.fc0.lin.weight"] = (h, c)
        shp["fc_center.fc0.act.lin_dir.weight"] = (h, h)
        shp["fc_center.lin1.weight"] = (1, h)
        shp["fc_center.act2.lin_dir.weight"] = (1, 1)
        shp["fc_center.shortcut.weight"] = (1, c)
    return shp


def make_encoder_weights(cfg, seed=0):
    """kaiming_uniform(a=sqrt(5)) bound 1/sqrt(fan_in), as VecLinear.reset_parameters (vec_layers.py:114-119)."""
    return {k: _uniform(seed, "enc." + k, s, 1.0 / np.sqrt(s[1])) for k, s in encoder_param_shapes(cfg).items()}


def decoder_layer_dims(cfg):
    """(in, out) per linear layer of DeepSDF_Decoder (deepsdf_decoder.py:33-57)."""
    dims = [cfg["latent_size"] + cfg["pe_dim"]] + list(cfg["dims"]) + [1]
    out = []
    for layer in range(len(dims) - 1):
        o = dims[layer + 1] - dims[0] if (layer + 1) in cfg["latent_in"] else dims[layer + 1]
        out.append((dims[layer], o))
    return out


def make_decoder_weights(cfg, seed=0):
    w = {}
    for layer, (i, o) in enumerate(decoder_layer_dims(cfg)):
        b = 1.0 / np.sqrt(i)
        w[f"lin{layer}.bias"] = _uniform(seed, f"dec.lin{layer}.bias", (o,), b)
        if cfg["weight_norm"] and layer in cfg["norm_layers"]:
            v = _uniform(seed, f"dec.lin{layer}.weight_v", (o, i), b)
            g = v.norm(dim=1, keepdim=True) * (1.0 + 0.2 * _uniform(seed, f"dec.lin{layer}.g", (o, 1), 1.0))
            w[f"lin{layer}.weight_g"], w[f"lin{layer}.weight_v"] = g, v
        else:
            w[f"lin{layer}.weight"] = _uniform(seed, f"dec.lin{layer}.weight", (o, i), b)
    return w


def to_checkpoint(enc_w, dec_w, epoch=0):
    """The on-disk format Shape_Prior.__init__ expects (model_utils.py:118-128)."""
    sd = {"network_dict.encoder." + k: v for k, v in enc_w.items()}
    sd.update({"network_dict.decoder." + k: v for k, v in dec_w.items()})
    return {"epoch": epoch, "model_state_dict": sd}


# ----------------------------------------------------------------------------- inputs
def _rand_rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def canonical_shape(N, seed):
    """A chair-like union of boxes sampled on its surfaces/volume: seat slab + back slab + 4 legs.
    Per-seed proportions make instances distinguishable (needed for the matcher tests)."""
    rng = _rng(seed, "shape")
    pr = rng.random(6)
    w, d, h = 0.35 + 0.15 * pr[0], 0.35 + 0.15 * pr[1], 0.35 + 0.25 * pr[2]
    seat_t, back_h, leg_w = 0.04 + 0.04 * pr[3], 0.3 + 0.3 * pr[4], 0.03 + 0.03 * pr[5]
    boxes = [((-w, w), (-d, d), (0.0, seat_t)),
             ((-w, w), (d - 0.05, d), (seat_t, seat_t + back_h))]
    for sx in (-1, 1):
        for sy in (-1, 1):
            cx, cy = sx * (w - leg_w), sy * (d - leg_w)
            boxes.append(((cx - leg_w, cx + leg_w), (cy - leg_w, cy + leg_w), (-h, 0.0)))
    vol = np.array([(b[0][1] - b[0][0]) * (b[1][1] - b[1][0]) * (b[2][1] - b[2][0]) + 1e-3 for b in boxes])
    which = rng.choice(len(boxes), size=N, p=vol / vol.sum())
    u = rng.random((N, 3))
    lo = np.array([[b[a][0] for a in range(3)] for b in boxes])[which]
    hi = np.array([[b[a][1] for a in range(3)] for b in boxes])[which]
    return (lo + u * (hi - lo)).astype(np.float64)


def make_instances(B, N=1024, seed=0, rigid=True):
    """cfg-2 style batch: B instances [B,3,N] float32, each a seeded shape under a random
    similarity transform (uniform SO(3), s in [0.5,1.5], t in U(-2,2)^3; mirrors vec_dgcnn_atten.py:280-286)."""
    out = np.empty((B, 3, N), dtype=np.float32)
    for b in range(B):
        p = canonical_shape(N, seed * 100003 + b)
        if rigid:
            rng = _rng(seed * 100003 + b, "pose")
            R, s, t = _rand_rot(rng), 0.5 + rng.random(), 4.0 * rng.random(3) - 2.0
            p = s * p @ R.T + t
        out[b] = p.T.astype(np.float32)
    return torch.from_numpy(out)


def make_scene_pair(n_obj=32, N=1024, seed=0, noise=0.005):
    """cfg-3 style FlyingShape-like pair (SURVEY.md 8d / eval_flyingshape.py:76,119-129): reference scene and
    rescan hold the same n_obj shapes; each rescan object is re-sampled, re-posed by an independent rigid
    transform and jittered.  Returns dict(ref [n,N,3], rescan [n,N,3], ref_T [n,4,4], rescan_T [n,4,4]);
    ground-truth match is the identity permutation, GT transform = rescan_T @ inv(ref_T)."""
    ref = np.empty((n_obj, N, 3), dtype=np.float32)
    res = np.empty((n_obj, N, 3), dtype=np.float32)
    ref_T = np.tile(np.eye(4), (n_obj, 1, 1))
    res_T = np.tile(np.eye(4), (n_obj, 1, 1))
    for i in range(n_obj):
        sid = seed * 100003 + i
        rng = _rng(sid, "scene")
        for arr, T, tag in ((ref, ref_T, 0), (res, res_T, 1)):
            p = canonical_shape(N, sid) if tag == 0 else canonical_shape(N, sid)[rng.permutation(N)]
            R, t = _rand_rot(rng), 4.0 * rng.random(3) - 2.0
            q = p @ R.T + t
            if tag == 1 and noise > 0:
                q = q + noise * rng.standard_normal(q.shape)
            arr[i] = q.astype(np.float32)
            T[i, :3, :3], T[i, :3, 3] = R, t
    return dict(ref=torch.from_numpy(ref), rescan=torch.from_numpy(res),
                ref_T=torch.from_numpy(ref_T.astype(np.float32)), rescan_T=torch.from_numpy(res_T.astype(np.float32)))


def make_queries(B, M, seed=0, box=1.1):
    """SDF query points ~ U(-box/2, box/2)^3 (MISE box size 1.1, mesh_extractor2.py:100)."""
    r = _rng(seed, "queries").random((B, M, 3))
    return torch.from_numpy(((r - 0.5) * box).astype(np.float32))


def make_raw_scan(shapes, seed, pmin=1024, pmax=60000, device=None):
    """A 3RScan-like scan for configs[3]: one raw cloud of pmin .. pmax points per instance (re-sampled canonical shape `shapes[i]` under
    a random rigid motion), zero-padded to the largest with a validity mask -- the dict Shape_Prior.encode_fps / More_Solver._solve_end2end
    take (eval_3rscan.py:78-95 heterogeneous batching).  -> ({'pc' [n,3,Pmax], 'pc_mask' [n,1,Pmax]}, total points)."""
    r = np.random.default_rng(seed)
    clouds = []
    for s_ in shapes:
        P = int(np.exp(r.uniform(np.log(pmin), np.log(pmax))))
        c = torch.as_tensor(canonical_shape(P, int(s_)), dtype=torch.float32)
        Rm = torch.as_tensor(_rand_rot(r), dtype=torch.float32)
        clouds.append(c @ Rm.T + torch.as_tensor(r.uniform(-2, 2, 3), dtype=torch.float32))
    mx = max(c.shape[0] for c in clouds)
    pc = torch.zeros(len(clouds), 3, mx)
    mask = torch.zeros(len(clouds), 1, mx, dtype=torch.bool)
    for i, c in enumerate(clouds):
        pc[i, :, :c.shape[0]] = c.T
        mask[i, :, :c.shape[0]] = True
    if device is not None:
        pc, mask = pc.to(device), mask.to(device)
    return {"pc": pc, "pc_mask": mask}, sum(c.shape[0] for c in clouds)
