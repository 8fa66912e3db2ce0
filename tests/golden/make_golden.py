#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING the reference's own Python modules
from /root/reference (dev container only -- the reference never travels; the fixtures are data).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Recipe: leaf files are loaded by path; the un-vendored third-party packages the
reference imports are replaced in sys.modules by
  * pytorch3d.ops.{knn_points, sample_farthest_points}  -> oracle.canon (canonical semantics, contract=0)
  * pycg / trimesh / mesh_extractor2                     -> empty stubs (GUI / meshing, not on the path)
Weights are the build's deterministic synthetic weights (livingscenes_amd.synth); the released checkpoint
is absent (/root/reference/.MISSING_LARGE_BLOBS).  Nothing from the reference's source text is stored.
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch
import yaml

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)

from livingscenes_amd import synth  # noqa: E402
from oracle import canon  # noqa: E402

CAPTURE = {"knn": [], "fps": []}


def _knn_points(p1, p2, K=1, return_nn=False, **kw):
    """pytorch3d.ops.knn_points shim: p1 [B,N1,D], p2 [B,N2,D], D = 3C with j = c*3+x."""
    B, N1, D = p1.shape
    C = D // 3
    d = p1.detach().reshape(B, N1, C, 3).permute(0, 1, 3, 2).contiguous().float().numpy()
    s = p2.detach().reshape(B, p2.shape[1], C, 3).permute(0, 1, 3, 2).contiguous().float().numpy()
    idx_np, dist_np = canon.knn_c(d, s, K, contract=0, return_dist=True)
    idx = torch.from_numpy(idx_np.astype(np.int64))
    CAPTURE["knn"].append(idx.clone())
    nn = None
    if return_nn:
        nn = torch.gather(p2[:, None].expand(-1, N1, -1, -1), 2, idx[..., None].expand(-1, -1, -1, D))
    return torch.from_numpy(dist_np).to(p1.dtype), idx, nn


def _sample_farthest_points(points, lengths=None, K=50, random_start_point=False):
    assert not random_start_point
    idx = torch.from_numpy(canon.fps_c(points.detach().float().contiguous().numpy(), K).astype(np.int64))
    CAPTURE["fps"].append(idx.clone())
    return torch.gather(points, 1, idx[..., None].expand(-1, -1, 3)), idx


def install_stubs():
    p3d = types.ModuleType("pytorch3d")
    ops = types.ModuleType("pytorch3d.ops")
    knn = types.ModuleType("pytorch3d.ops.knn")
    pa = types.ModuleType("pytorch3d.ops.points_alignment")
    knn.knn_points = _knn_points
    ops.knn_points = _knn_points
    ops.sample_farthest_points = _sample_farthest_points
    ops.knn = knn
    pa.iterative_closest_point = None
    pa.SimilarityTransform = None
    p3d.ops = ops
    sys.modules.update({"pytorch3d": p3d, "pytorch3d.ops": ops, "pytorch3d.ops.knn": knn,
                        "pytorch3d.ops.points_alignment": pa})
    pycg = types.ModuleType("pycg")
    pycg.vis = pycg.image = pycg.exp = None
    sys.modules["pycg"] = pycg
    sys.modules["trimesh"] = types.ModuleType("trimesh")


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def t2n(d):
    return {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def main():
    torch.set_num_threads(8)
    install_stubs()
    VS = "lib_shape_prior/core/lib/vec_sim3/"
    vl = load_by_path("vec_layers", VS + "vec_layers.py")
    att = load_by_path("ref_vec_dgcnn_atten", VS + "vec_dgcnn_atten.py")
    dsdf = load_by_path("ref_deepsdf_decoder", "lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py")
    sys.path.insert(0, REF)
    from lib_math import torch_se3  # noqa
    from lib_more import matcher_new, pose_estimation  # noqa
    act = torch.nn.LeakyReLU(negative_slope=0.2, inplace=False)

    # ---------------------------------------------------------------- 1. VN layer known-answer vectors
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(4321)   # the modules below draw their initial weights from the GLOBAL generator: seeded -> reproducible fixture
    out = {}
    x = torch.randn(2, 6, 3, 5, 4, generator=g)
    lin = vl.VecLinear(6, 8, mode="so3")
    lna = vl.VecLinearNormalizeActivate(6, 8, mode="so3", act_func=act)
    lna_sh = vl.VecLinearNormalizeActivate(6, 8, mode="so3", act_func=act, shared_nonlinearity=True)
    rb = vl.VecResBlock(6, 1, 4, act_func=act, mode="so3")
    with torch.no_grad():
        out["x"] = x
        out["lin_w"], out["lin_y"] = lin.weight, lin(x)
        out["lna_w"], out["lna_wd"], out["lna_y"] = lna.lin.weight, lna.act.lin_dir.weight, lna(x)
        out["lnash_w"], out["lnash_wd"], out["lnash_y"] = lna_sh.lin.weight, lna_sh.act.lin_dir.weight, lna_sh(x)
        out["cevn_y"] = vl.channel_equi_vec_normalize(x)
        x4 = x[..., 0]
        out["rb_y"] = rb(x4)
        for k, v in rb.state_dict().items():
            out["rb." + k] = v
    np.savez_compressed(os.path.join(HERE, "vn_layers.npz"), **t2n(out))

    # ---------------------------------------------------------------- 2. small-config encoder, full trace
    cfg_s = synth.small_encoder_cfg()
    w_s = synth.make_encoder_weights(cfg_s, seed=7)
    net = att.VecDGCNN_att(**cfg_s).eval()
    net.load_state_dict(w_s, strict=True)
    xs = synth.make_instances(2, 128, seed=3, rigid=False)
    xs = xs - xs.mean(-1, keepdim=True)
    CAPTURE["knn"].clear(), CAPTURE["fps"].clear()
    with torch.no_grad():
        center, scale, z_so3, z_inv = net(xs)
    out = {"x": xs, "center": center, "scale": scale, "z_so3": z_so3, "z_inv": z_inv}
    for i, t in enumerate(CAPTURE["knn"]):
        out[f"knn_idx_{i}"] = t.to(torch.int32)
    for i, t in enumerate(CAPTURE["fps"]):
        out[f"fps_idx_{i}"] = t.to(torch.int32)
    np.savez_compressed(os.path.join(HERE, "encoder_small.npz"), **t2n(out))

    # ---------------------------------------------------------------- 2b. the same forward, PER LAYER (hooks on the reference's own
    # modules; nothing is recomputed here): fixture of the isolated operator tests (ls_vn_edgeconv_* / ls_vn_lna_f32 /
    # ls_encoder_tail_f32).  V_list[i] receives the graph feature y = cat(nbr - ctr, ctr) [B,2C,3,Nd,K] (vec_dgcnn_atten.py:160):
    # its second channel half at k = 0 is the layer's destination feature; global_conv_list[j] receives cat(msg, mean) and returns
    # the layer output (:222-225).  Stored in the library's row layout [B,N,3,C].
    cap = {}

    def rows(f):  # [B,C,3,N] -> [B,N,3,C]
        return f.permute(0, 3, 2, 1).contiguous()
    hooks = []
    ggf, layer_no = net.get_graph_feature, [0]

    def record_graph_inputs(*a, **kw):   # the reference's own method, arguments recorded on the way in (:196-199)
        cap[f"src_{layer_no[0]}"], cap[f"dst_in_{layer_no[0]}"] = rows(kw["src_f"]), rows(kw["dst_f"])
        layer_no[0] += 1
        return ggf(*a, **kw)
    net.get_graph_feature = record_graph_inputs
    for j in range(len(net.global_conv_list)):
        i = j + cfg_s["res_global_start_layer"]

        def pre_g(mod, args, i=i):
            x_ = args[0]
            cap[f"msg_{i}"] = rows(x_[:, : x_.shape[1] // 2])

        def post_g(mod, args, outp, i=i):
            cap[f"out_{i}"] = rows(outp)
        hooks.append(net.global_conv_list[j].register_forward_pre_hook(pre_g))
        hooks.append(net.global_conv_list[j].register_forward_hook(post_g))
    CAPTURE["knn"].clear(), CAPTURE["fps"].clear()
    with torch.no_grad():
        center2, scale2, z_so3_2, z_inv_2 = net(xs)
    for h in hooks:
        h.remove()
    del net.get_graph_feature
    assert torch.equal(z_so3_2, z_so3) and torch.equal(z_inv_2, z_inv)
    # layers without a global conv hand their message straight to the next layer as its source features
    # (not stored twice: msg_i == out_i == src_{i+1} for i < res_global_start_layer)
    layers = {"x": xs, "center": center2, "scale": scale2, "z_so3": z_so3_2, "z_inv": z_inv_2}
    layers.update(cap)
    for i, t in enumerate(CAPTURE["knn"]):
        layers[f"knn_idx_{i}"] = t.to(torch.int32)
    for i, t in enumerate(CAPTURE["fps"]):
        layers[f"fps_idx_{i}"] = t.to(torch.int32)
    np.savez_compressed(os.path.join(HERE, "encoder_small_layers.npz"), **t2n(layers))

    # ---------------------------------------------------------------- 3. released config through the UNMODIFIED model_utils.Shape_Prior
    for dotted in ["lib_shape_prior", "lib_shape_prior.core", "lib_shape_prior.core.lib",
                   "lib_shape_prior.core.lib.implicit_func", "lib_shape_prior.core.lib.vec_sim3",
                   "lib_shape_prior.core.models", "lib_shape_prior.core.models.utils",
                   "lib_shape_prior.core.models.utils.occnet_utils"]:
        m = types.ModuleType(dotted)
        m.__path__ = []
        sys.modules[dotted] = m
    IF = "lib_shape_prior/core/lib/implicit_func/"
    load_by_path("lib_shape_prior.core.lib.implicit_func.onet_decoder", IF + "onet_decoder.py")
    sys.modules["lib_shape_prior.core.lib.implicit_func.deepsdf_decoder"] = dsdf
    load_by_path("lib_shape_prior.core.lib.vec_sim3.vec_dgcnn", VS + "vec_dgcnn.py")
    sys.modules["lib_shape_prior.core.lib.vec_sim3.vec_dgcnn_atten"] = att
    load_by_path("lib_shape_prior.core.lib.vec_sim3.pcnet", VS + "pcnet.py")
    me = types.ModuleType("lib_shape_prior.core.models.utils.occnet_utils.mesh_extractor2")
    me.Generator3D = object
    sys.modules[me.__name__] = me
    import model_utils  # the reference's file, unmodified

    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    with open(os.path.join(REF, "weights/files_backup/model_config.yaml")) as f:
        ref_cfg = yaml.full_load(f)
    assert {k: ref_cfg["model"]["encoder"][k] for k in ecfg} == ecfg, "synth encoder cfg != released config"
    assert {k: ref_cfg["model"]["decoder"][k] for k in dcfg} == dcfg, "synth decoder cfg != released config"
    enc_w, dec_w = synth.make_encoder_weights(ecfg, seed=0), synth.make_decoder_weights(dcfg, seed=0)
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "ckpt.pt")
        torch.save(synth.to_checkpoint(enc_w, dec_w), ck)
        sp = model_utils.Shape_Prior(
            {"working_dir": "/", "field_cfg": os.path.join(REF, "weights/files_backup/model_config.yaml"),
             "field_pt": ck}, "chair", use_double=False).eval()
    xi = synth.make_instances(2, 1024, seed=0)
    CAPTURE["knn"].clear(), CAPTURE["fps"].clear()
    with torch.no_grad():
        emb = sp.encode(xi)
        q = synth.make_queries(2, 256, seed=0) * emb["s"][:, None, None] + emb["t"]
        sdf = sp.decoder(q, None, emb, return_sdf=True)
    out = {"z_so3": emb["z_so3"], "z_inv": emb["z_inv"], "s": emb["s"], "t": emb["t"], "query": q, "sdf": sdf}
    for i, t in enumerate(CAPTURE["knn"]):
        out[f"knn_idx_{i}"] = t.to(torch.int16)
    for i, t in enumerate(CAPTURE["fps"]):
        out[f"fps_idx_{i}"] = t.to(torch.int16)
    np.savez_compressed(os.path.join(HERE, "shape_prior_full.npz"), **t2n(out))

    # ---------------------------------------------------------------- 4. matchers
    out = {}
    g = torch.Generator().manual_seed(99)
    cases = {"n1": (1, 1), "n2": (2, 3), "n3": (3, 3), "n5": (5, 4), "n32": (32, 32)}
    for name, (n, m) in cases.items():
        a, b = torch.randn(n, 256, generator=g), torch.randn(m, 256, generator=g)
        if name == "n32":  # realistic: rescan codes = permuted noisy ref codes
            b = a[torch.randperm(n, generator=g)] + 0.3 * torch.randn(n, 256, generator=g)
        r = matcher_new.sequential_matcher(a, b)
        out[f"seq_{name}_a"], out[f"seq_{name}_b"] = a, b
        out[f"seq_{name}_m0"], out[f"seq_{name}_m1"] = r["matches0"], r["matches1"]
    # all-negative scores (renormalisation by a negative max flips the order, matcher_new.py:123)
    a = torch.rand(4, 256, generator=g) + 0.1
    b = -(torch.rand(3, 256, generator=g) + 0.1)
    r = matcher_new.sequential_matcher(a, b)
    out["seq_neg_a"], out["seq_neg_b"], out["seq_neg_m0"], out["seq_neg_m1"] = a, b, r["matches0"], r["matches1"]
    # exact ties: duplicated rows/cols
    a = torch.randn(3, 256, generator=g)
    a = torch.cat([a, a[:1]], 0)
    b = torch.cat([a[1:2], a[:1], a[:1]], 0)
    r = matcher_new.sequential_matcher(a, b)
    out["seq_tie_a"], out["seq_tie_b"], out["seq_tie_m0"], out["seq_tie_m1"] = a, b, r["matches0"], r["matches1"]
    a, b = torch.randn(6, 256, generator=g), torch.randn(7, 256, generator=g)
    r = matcher_new.nn_matcher(a.T[None], b.T[None])
    out["nn_a"], out["nn_b"], out["nn_m0"], out["nn_m1"] = a, b, r["matches0"], r["matches1"]
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a_, **k_: self  # sinkhorn_matcher hard-codes .cuda() (matcher_new.py:53)
    try:
        r = matcher_new.sinkhorn_matcher(a.T[None], b.T[None])
    finally:
        torch.Tensor.cuda = orig_cuda
    out["sk_m0"], out["sk_m1"] = r["matches0"], r["matches1"]
    src = {"z_inv": torch.randn(5, 256, generator=g), "z_so3": torch.randn(5, 256, 3, generator=g)}
    Rp = torch.linalg.qr(torch.randn(5, 3, 3, generator=g))[0]
    perm = torch.randperm(5, generator=g)
    tgt = {"z_inv": src["z_inv"][perm] + 0.1 * torch.randn(5, 256, generator=g),
           "z_so3": torch.einsum("nij,ncj->nci", Rp, src["z_so3"][perm]) + 0.05 * torch.randn(5, 256, 3, generator=g)}
    for nm, fn in (("eq", matcher_new.eq_seq_matcher), ("sim3", matcher_new.sim3_seq_matcher)):
        r = fn(src, tgt)
        out[f"{nm}_m0"], out[f"{nm}_m1"] = r["matches0"], r["matches1"]
    out["eqsrc_z_inv"], out["eqsrc_z_so3"], out["eqtgt_z_inv"], out["eqtgt_z_so3"] = \
        src["z_inv"], src["z_so3"], tgt["z_inv"], tgt["z_so3"]
    np.savez_compressed(os.path.join(HERE, "matchers.npz"), **t2n(out))

    # ---------------------------------------------------------------- 5. Kabsch, SE(3), metrics
    out = {}
    b, n = 32, 256
    x1 = torch.randn(b, n, 3, generator=g)
    Rg = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0]
    Rg[:16] = Rg[:16] * torch.sign(torch.det(Rg[:16]))[:, None, None]          # proper rotations
    Rg[16:] = Rg[16:] * (-torch.sign(torch.det(Rg[16:])))[:, None, None]       # reflections (det<0 branch)
    tg = torch.randn(b, 3, 1, generator=g)
    x2 = (Rg @ x1.transpose(1, 2) + tg).transpose(1, 2) + 0.01 * torch.randn(b, n, 3, generator=g)
    wts = torch.rand(b, n, generator=g)
    R, t, res, flag = pose_estimation.kabsch_transformation_estimation(x1, x2)
    Rw, tw, resw, _ = pose_estimation.kabsch_transformation_estimation(x1, x2, wts)
    out.update(kab_x1=x1, kab_x2=x2, kab_w=wts, kab_R=R, kab_t=t, kab_res=res, kab_Rw=Rw, kab_tw=tw, kab_resw=resw)
    out["rot_err"] = pose_estimation.rotation_error(R, Rg)
    out["kab_Rg"], out["kab_tg"] = Rg, tg
    out["trans_err"] = pose_estimation.translation_error(t, tg)
    T1, T2 = torch_se3.Rt_to_SE3(R, t), torch_se3.Rt_to_SE3(Rg, tg)
    out["se3_T1"], out["se3_T2"] = T1, T2
    out["se3_inv"] = torch_se3.inverse(T1)
    out["se3_cat"] = torch_se3.concatenate(T1, T2)
    out["se3_tf"] = torch_se3.transform(T1, x1)
    out["rmse"] = pose_estimation.compute_transformation_error(x1[:1], x2[:1], T1[:1], T2[:1])
    out["inv3d"] = pose_estimation.inverse_3d_transform(T1)
    np.savez_compressed(os.path.join(HERE, "registration.npz"), **t2n(out))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
