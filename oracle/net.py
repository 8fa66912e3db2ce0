"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU restatement of the LivingScenes encoder / decoder.

Same op order as the reference modules (all paths relative to /root/reference), so that on
identical weights and inputs it reproduces them to fp32 round-off.  It is pinned against the
*imported* reference modules by tests/golden/make_golden.py (run in the dev container, where
/root/reference exists); the resulting fixtures are what tests/test_oracle_golden.py checks on
any machine.  The k-NN / FPS leaves come from oracle.canon (pytorch3d is absent: parity for
those two ops is UNPINNED, see canon.py).

Functional style: ``w`` is a dict of tensors keyed by the reference's state_dict names
(``V_list.0.lin.weight`` ...; decoder ``lin0.weight_g`` ...), ``cfg`` the encoder/decoder kwargs
of weights/files_backup/model_config.yaml.  Only tests/, smoke() and bench.py's cpu_baseline
leg may import this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import canon

KNN_CONTRACT = 0  # canonical arithmetic mode used by the oracle graph (see ls_oracle.c)


# ----------------------------------------------------------------------------- VN layers
def vec_linear(v, W):
    """VecLinear.forward, so3 mode, no scalar path: lib_shape_prior/core/lib/vec_sim3/vec_layers.py:121-136."""
    return F.linear(v.transpose(1, -1), W).transpose(-1, 1)


def vec_activation(x, W_dir, neg_slope):
    """VecActivation.forward, so3 mode, no normalization: vec_layers.py:241-268."""
    k = vec_linear(x, W_dir)
    k_dir = F.normalize(k, dim=2)
    q_para_len = (x * k_dir).sum(dim=2, keepdim=True)
    q_orthogonal = x - q_para_len * k_dir
    acted_len = F.leaky_relu(q_para_len, negative_slope=neg_slope)
    return q_orthogonal + k_dir * acted_len


def vec_lna(x, W, W_dir, neg_slope):
    """VecLinearNormalizeActivate.forward (vector only): vec_layers.py:523-534."""
    return vec_activation(vec_linear(x, W), W_dir, neg_slope)


def channel_equi_vec_normalize(x):
    """vec_layers.py:24-31."""
    x_dir = F.normalize(x, dim=2)
    x_norm = x.norm(dim=2, keepdim=True)
    x_normalized_norm = F.normalize(x_norm, dim=1)
    return x_dir * x_normalized_norm


def vec_resblock(v, w, prefix, neg_slope):
    """VecResBlock.forward, so3, no scalar path, last_activate, in!=out: vec_layers.py:631-651."""
    v_net = vec_lna(v, w[prefix + "fc0.lin.weight"], w[prefix + "fc0.act.lin_dir.weight"], neg_slope)
    dv = vec_linear(v_net, w[prefix + "lin1.weight"])
    v_s = vec_linear(v, w[prefix + "shortcut.weight"])
    return vec_activation(v_s + dv, w[prefix + "act2.lin_dir.weight"], neg_slope)


# ----------------------------------------------------------------------------- pytorch3d leaves
def _to_rows(f):
    """[B,C,3,N] (reference layout) -> [B,N,3,C] x-major rows used by oracle.canon."""
    return f.permute(0, 3, 2, 1).contiguous().numpy()


def knn_points(dst_flat, src_flat, K, C, idx=None):
    """pytorch3d.ops.knn_points(dst [B,Nd,3C], src [B,Ns,3C], K, return_nn=True) as called at
    vec_dgcnn_atten.py:139-141; feature index j = c*3+x (reshape of [B,C,3,N] at :138).
    ``idx`` (tests only): neighbour lists to use instead of searching -- see encoder_forward(graph=...)."""
    B, Nd, D = dst_flat.shape
    Ns = src_flat.shape[1]
    if idx is None:
        d = dst_flat.reshape(B, Nd, C, 3).permute(0, 1, 3, 2).contiguous().numpy()
        s = src_flat.reshape(B, Ns, C, 3).permute(0, 1, 3, 2).contiguous().numpy()
        idx = torch.from_numpy(canon.knn_c(d, s, K, contract=KNN_CONTRACT).astype(np.int64))
    else:
        idx = torch.as_tensor(idx).to(torch.int64).reshape(B, Nd, K)
    nn = torch.gather(src_flat[:, None].expand(-1, Nd, -1, -1), 2, idx[..., None].expand(-1, -1, -1, D))
    return None, idx, nn


def sample_farthest_points(points, K, lengths=None):
    """pytorch3d.ops.sample_farthest_points(points [B,N,3], K, random_start_point=False)."""
    idx = torch.from_numpy(
        canon.fps_c(points.contiguous().numpy(), K, lengths=lengths, contract=KNN_CONTRACT).astype(np.int64))
    pts = torch.gather(points, 1, idx.clamp(min=0)[..., None].expand(-1, -1, 3))
    return pts, idx


# ----------------------------------------------------------------------------- encoder
def get_graph_feature(src_f, dst_f, k, cross, idx=None):
    """VecDGCNN_att.get_graph_feature, use_dg branch: vec_dgcnn_atten.py:124-161.
    Conscious divergence: the reference calls torch.cross WITHOUT dim (:157), which crosses over the
    batch axis when B == 3; the restatement (and the HIP path) always cross over xyz (dim=2)."""
    B, C, _, N_src = src_f.shape
    N_dst = dst_f.shape[-1]
    _dst, _src = dst_f.reshape(B, -1, N_dst), src_f.reshape(B, -1, N_src)
    _, knn_idx, nn = knn_points(_dst.transpose(2, 1), _src.transpose(2, 1), k, C, idx=idx)
    nn = nn.reshape(B, N_dst, k, C, 3).permute(0, -2, -1, 1, 2)
    dst_pad = dst_f[..., None].expand_as(nn)
    if cross:
        x_dir = F.normalize(src_f, dim=2)
        x_dir_pad = x_dir[..., None].expand_as(nn)
        cr = torch.cross(x_dir_pad, nn, dim=2)
        y = torch.cat([cr, nn - dst_pad, dst_pad], 1)
    else:
        y = torch.cat([nn - dst_pad, dst_pad], 1)
    return y, knn_idx


def down_sample(x, f, factor):
    """VecDGCNN_att.down_sample: vec_dgcnn_atten.py:163-175."""
    N_new = x.shape[-1] // factor
    x_new, idx = sample_farthest_points(x.squeeze(1).transpose(2, 1), N_new)
    x_new = x_new.transpose(2, 1).unsqueeze(1).type(x.dtype)
    C = f.shape[1]
    f_new = torch.gather(f, dim=-1, index=idx[:, None, None, :].expand(-1, C, 3, -1))
    return x_new, f_new, idx


def as_params(w):
    """The reference holds its weights as nn.Parameters (requires_grad=True) and runs under no_grad();
    ATen's matmul picks its folding strategy from ``requires_grad`` (so the GEMM shapes, hence the fp32
    rounding, depend on it).  Mirror that so the restatement is bit-comparable with the reference."""
    return {k: (v if v.requires_grad else v.detach().clone().requires_grad_(True)) for k, v in w.items()}


@torch.no_grad()
def encoder_forward(w, cfg, x, trace=None, graph=None):
    """VecDGCNN_att.forward: vec_dgcnn_atten.py:177-252.  x [B,3,N] -> (center, scale, z_so3, z_inv).
    If ``trace`` is a dict it receives per-layer fps idx / knn idx / layer inputs / outputs.
    ``graph`` (tests only): {layer: neighbour lists [B,Nd,K]} to use instead of the k-NN search of that layer.  The deeper layers
    search in FEATURE space, where two candidates can be equidistant to within fp32 round-off; which of them enters the list then
    depends on the summation order of the GEMMs that produced the features, on any implementation (the reference's own CPU and CUDA
    paths differ the same way).  A test that finds such a flip re-runs the restatement on the device's lists and compares the codes
    at full tolerance, after checking that every flipped pair IS a near-tie."""
    w = as_params(w)
    ns = cfg.get("leak_neg_slope", 0.2)
    L = cfg["num_layers"]
    hc = cfg["atten_multi_head_c"]
    ds_layers, ds_factor = cfg["down_sample_layers"], cfg["down_sample_factor"]
    a0, g0 = cfg["atten_start_layer"], cfg["res_global_start_layer"]
    K = cfg["num_knn"]
    src_xyz, src_f = x.unsqueeze(1), x.unsqueeze(1)
    for i in range(L):
        if i in ds_layers:
            dst_xyz, dst_f, fidx = down_sample(src_xyz, src_f, ds_factor[ds_layers.index(i)])
            if trace is not None:
                trace[f"fps_idx_{i}"] = fidx
        else:
            dst_xyz, dst_f = src_xyz, src_f
        if trace is not None:
            trace[f"src_f_{i}"], trace[f"dst_f_in_{i}"] = src_f, dst_f
        y, knn_idx = get_graph_feature(src_f, dst_f, K, cross=(i == 0), idx=None if graph is None else graph.get(i))
        if trace is not None:
            trace[f"knn_idx_{i}"] = knn_idx
        Wv, Wvd = w[f"V_list.{i}.lin.weight"], w[f"V_list.{i}.act.lin_dir.weight"]
        if i < a0:
            dst_f = vec_lna(y, Wv, Wvd, ns).mean(dim=-1)
        else:
            k = vec_lna(y, w[f"K_list.{i}.lin.weight"], w[f"K_list.{i}.act.lin_dir.weight"], ns)
            q = vec_lna(dst_f, w[f"Q_list.{i}.lin.weight"], w[f"Q_list.{i}.act.lin_dir.weight"], ns)
            v = vec_lna(y, Wv, Wvd, ns)
            k = channel_equi_vec_normalize(k)
            q = channel_equi_vec_normalize(q)
            qk = (k * q[..., None]).sum(2)
            B, C, N, Kn = qk.shape
            n_head = C // hc
            qk_c = qk.reshape(B, n_head, hc, N, Kn)
            atten = qk_c.sum(2, keepdim=True) / np.sqrt(3 * hc)
            atten = torch.softmax(atten, dim=-1)
            atten = atten.expand(-1, -1, hc, -1, -1).reshape(qk.shape).unsqueeze(2)
            dst_f = (atten * v).sum(-1)
        if trace is not None:
            trace[f"msg_f_{i}"] = dst_f
        if cfg.get("use_res_global_conv", True) and i >= g0:
            g = dst_f.mean(dim=-1)
            dst_f = torch.cat([dst_f, g[..., None].expand_as(dst_f)], 1)
            j = i - g0
            dst_f = vec_lna(dst_f, w[f"global_conv_list.{j}.lin.weight"],
                            w[f"global_conv_list.{j}.act.lin_dir.weight"], ns)
        if trace is not None:
            trace[f"dst_f_{i}"] = dst_f
        src_xyz, src_f = dst_xyz, dst_f

    xx = vec_lna(dst_f, w["conv_c.lin.weight"], w["conv_c.act.lin_dir.weight"], ns)
    xx = xx.mean(dim=-1)
    z_so3 = channel_equi_vec_normalize(xx)
    scale = xx.norm(dim=-1).mean(1) * cfg["scale_factor"]
    z_inv_dual = vec_linear(xx[..., None], w["fc_inv.weight"]).squeeze(-1)
    z_inv = (channel_equi_vec_normalize(z_inv_dual) * z_so3).sum(-1)
    center = vec_resblock(xx[..., None], w, "fc_center.", ns).squeeze(-1)
    if cfg.get("center_pred_scale", False):
        center = center * cfg["scale_factor"]
    return center, scale, z_so3, z_inv


@torch.no_grad()
def shape_prior_encode(w, cfg, x, trace=None):
    """Shape_Prior.encode: model_utils.py:165-197 (use_double=False, configs/room4cates.yaml:15)."""
    B = x.shape[0]
    centroid = x.mean(-1)
    pcl = x - centroid[..., None]
    dist = torch.cdist(pcl.transpose(-1, -2), pcl.transpose(-1, -2))
    scale_0 = dist.view(B, -1).topk(5, dim=-1)[0].mean(-1)
    pcl = pcl / scale_0[:, None, None]
    if trace is not None:
        trace["scale_0"], trace["pcl_norm"] = scale_0, pcl
    center_pred, pred_scale, z_so3, z_inv = encoder_forward(w, cfg, pcl, trace)
    centroid = center_pred.squeeze(1) + centroid
    return {"z_so3": z_so3, "z_inv": z_inv, "s": scale_0 * pred_scale, "t": centroid.unsqueeze(1)}


# ----------------------------------------------------------------------------- decoder
def fold_weight_norm(g, v):
    """torch.nn.utils.weight_norm (dim=0): W = g * v / ||v||_row  (deepsdf_decoder.py:52-57)."""
    return g * v / v.norm(dim=1, keepdim=True)


@torch.no_grad()
def decoder_forward(w, cfg, inp):
    """DeepSDF_Decoder.forward(input [B,M,L], 'val'): lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py:78-123."""
    w = as_params(w)
    dims = [cfg["latent_size"] + cfg["pe_dim"]] + list(cfg["dims"]) + [1]
    n_layers = len(dims)
    B, M, Lw = inp.shape
    x0 = inp.reshape(-1, Lw)
    x = x0
    for layer in range(n_layers - 1):
        if layer in cfg["latent_in"]:
            x = torch.cat([x, x0], 1)
        if cfg["weight_norm"] and layer in cfg["norm_layers"]:
            W = fold_weight_norm(w[f"lin{layer}.weight_g"], w[f"lin{layer}.weight_v"])
        else:
            W = w[f"lin{layer}.weight"]
        x = F.linear(x, W, w[f"lin{layer}.bias"])
        if layer < n_layers - 2:
            x = F.relu(x)
    return torch.tanh(x).view(B, M)


def field_query(w_dec, cfg_dec, query, code):
    """FieldWrapper.forward(query, None, code, return_sdf=True), decoder_type 'inner_deepsdf':
    model_utils.py:230-263."""
    B, M, _ = query.shape
    z_so3, z_inv = code["z_so3"], code["z_inv"]
    q = (query - code["t"]) / code["s"][:, None, None]
    inner = (q.unsqueeze(1) * z_so3.unsqueeze(2)).sum(dim=-1)
    length = q.norm(dim=-1).unsqueeze(1)
    inv_query = torch.cat([inner, length], 1).transpose(2, 1)
    inp = torch.cat([z_inv[:, None, :].expand(-1, M, -1), inv_query], -1)
    return decoder_forward(w_dec, cfg_dec, inp)


def field_query_with_grad(w_dec, cfg_dec, query, code):
    """field_query with autograd enabled (decoder_forward is wrapped in no_grad for inference): the reference's training-time
    graph through FieldWrapper.forward + DeepSDF_Decoder.forward, used as the oracle of ls_sdf_backward
    (more_solver.py:212-216: loss.backward() w.r.t. the code)."""
    B, M, _ = query.shape
    z_so3, z_inv = code["z_so3"], code["z_inv"]
    q = (query - code["t"]) / code["s"][:, None, None]
    inner = (q.unsqueeze(1) * z_so3.unsqueeze(2)).sum(dim=-1)
    length = q.norm(dim=-1).unsqueeze(1)
    inv_query = torch.cat([inner, length], 1).transpose(2, 1)
    inp = torch.cat([z_inv[:, None, :].expand(-1, M, -1), inv_query], -1)
    with torch.enable_grad():
        return decoder_forward.__wrapped__(w_dec, cfg_dec, inp)

