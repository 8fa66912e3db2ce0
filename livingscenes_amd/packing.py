"""Weight packing: reference state_dict -> the flat fp32 blob + ls_model_desc the HIP library consumes.

Done once at load time (Shape_Prior.__init__, /root/reference/model_utils.py:118-128 loads the same tensors).
All folds are computed in float64 and rounded once to fp32:

  * weight-norm          W = g * v / ||v||_row                       (deepsdf_decoder.py:52-57)
  * edge-conv tables     lin(E) = W1 src[nbr] + (W2-W1) dst ,  lin_dir(lin(E)) = Wd W1 src[nbr] + Wd (W2-W1) dst
                         (csrc/edge.hip header; W = [W1 | W2] acts on cat([nbr-ctr, ctr]), vec_dgcnn_atten.py:160)
  * global conv          W = [Wa | Wb] acts on cat([f, mean f]) (vec_dgcnn_atten.py:222-225)
  * decoder code layers  W u + b with u = [z_inv | <q,z_so3> | |q|]  ->  Wa^T, Wb^T, w_len  (csrc/sdf.hip header)

Blob layout (all row-major fp32, every tensor 4-float aligned):
  off_l0        [6][C0]            {W[:,0], W[:,1], W[:,2], (Wd W)[:,0], (Wd W)[:,1], (Wd W)[:,2]}   (layer 0: cross | nbr-ctr | ctr)
  off_edge[i]   [ncols][Cin]       pool: {V:W1, Wd W1, (W2-W1), Wd(W2-W1)} ; attn: {V:W1, WdW1, K:W1, WdW1, V:(W2-W1), Wd(..), K:(W2-W1), Wd(..), Q:Wq, Wdq Wq}
  off_glob[i]   [4C][C]            {Wa, Wd Wa, Wb, Wd Wb}
  off_convc     [Cdp][Cl]          rows 0..Cd-1 conv_c.lin, row Cd = lin_dir(1xCd) @ lin, zero padded to Cdp = ceil4(Cd+1)
  off_inv_t     [Cd][Cd]           fc_inv^T
  off_c_fc0_t   [Cd][2h]           {fc0.lin ; fc0.dir @ fc0.lin}^T
  off_c_misc    [h + Cd + 1]       lin1 | shortcut | act2.lin_dir scalar
  off_dec_w[l]  [out_l][kin_l]     folded decoder weights (layer latent_in-1 padded to ceil4 rows, latent_in: only the h part)
  off_dec_b[l]  [out_l]
  off_dec_inv_t/so3_t/len[l]       for l in {0, latent_in}: Wa^T [latent][out], Wb^T [latent][out], w_len [out]
"""
import numpy as np
import torch

from ._lib import LS_MAX_LAYERS, ModelDesc


def _f64(t):
    return t.detach().cpu().double().numpy()


class _Blob:
    def __init__(self):
        self.parts = []
        self.n = 0

    def add(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        off = self.n
        pad = (-arr.size) % 4
        self.parts.append(arr)
        if pad:
            self.parts.append(np.zeros(pad))
        self.n += arr.size + pad
        return off

    def finish(self):
        return np.concatenate(self.parts).astype(np.float32)


def pack_model(enc_w, enc_cfg, dec_w=None, dec_cfg=None):
    """-> (ModelDesc, blob float32 ndarray)."""
    d = ModelDesc()
    blob = _Blob()
    L = enc_cfg["num_layers"]
    assert L <= LS_MAX_LAYERS
    fd = list(enc_cfg["feat_dim"])
    d.num_layers = L
    for i in range(LS_MAX_LAYERS):
        d.feat_dim[i] = fd[i] if i < L else 0
        d.down_factor[i] = 1
    for lyr, fac in zip(enc_cfg["down_sample_layers"], enc_cfg["down_sample_factor"]):
        d.down_factor[lyr] = fac
    a0 = enc_cfg["atten_start_layer"]
    g0 = enc_cfg["res_global_start_layer"] if enc_cfg.get("use_res_global_conv", True) else L
    d.atten_start_layer, d.atten_head_c, d.res_global_start_layer = a0, enc_cfg["atten_multi_head_c"], g0
    d.num_knn, d.c_dim = enc_cfg["num_knn"], enc_cfg["c_dim"]
    d.center_pred = int(bool(enc_cfg.get("center_pred", False)))
    d.center_pred_scale = int(bool(enc_cfg.get("center_pred_scale", False)))
    d.scale_factor, d.neg_slope = float(enc_cfg["scale_factor"]), float(enc_cfg.get("leak_neg_slope", 0.2))
    assert enc_cfg.get("use_dg", True), "only the dynamic-graph (use_dg=True) encoder is implemented"
    assert not enc_cfg.get("z_so3_as_Omtx", False)

    def W(name):
        return _f64(enc_w[name])

    # layer 0: 3 input channels (cross, nbr-ctr, ctr)
    w0, wd0 = W("V_list.0.lin.weight"), W("V_list.0.act.lin_dir.weight")
    assert w0.shape[1] == 3
    dw = wd0 @ w0
    d.off_l0 = blob.add(np.stack([w0[:, 0], w0[:, 1], w0[:, 2], dw[:, 0], dw[:, 1], dw[:, 2]], 0))

    def edge_rows(prefix, cin):
        w, wd = W(prefix + ".lin.weight"), W(prefix + ".act.lin_dir.weight")
        w1, w2 = w[:, :cin], w[:, cin:]
        return [w1, wd @ w1], [w2 - w1, wd @ (w2 - w1)]

    for i in range(1, L):
        cin = fd[i - 1]
        pv, qv = edge_rows(f"V_list.{i}", cin)
        if i >= a0:
            pk, qk = edge_rows(f"K_list.{i}", cin)
            wq, wdq = W(f"Q_list.{i}.lin.weight"), W(f"Q_list.{i}.act.lin_dir.weight")
            rows = pv + pk + qv + qk + [wq, wdq @ wq]
        else:
            rows = pv + qv
        d.off_edge[i] = blob.add(np.concatenate(rows, 0))
    for i in range(L):
        if i >= g0:
            j = i - g0
            w, wd = W(f"global_conv_list.{j}.lin.weight"), W(f"global_conv_list.{j}.act.lin_dir.weight")
            c = fd[i]
            wa, wb = w[:, :c], w[:, c:]
            d.off_glob[i] = blob.add(np.concatenate([wa, wd @ wa, wb, wd @ wb], 0))
    cd = enc_cfg["c_dim"]
    wc, wdc = W("conv_c.lin.weight"), W("conv_c.act.lin_dir.weight")
    cdp = (cd + 1 + 3) // 4 * 4
    convc = np.zeros((cdp, fd[-1]))
    convc[:cd] = wc
    convc[cd] = (wdc @ wc)[0]
    d.off_convc = blob.add(convc)
    d.off_inv_t = blob.add(W("fc_inv.weight").T)
    h = cd // 2
    if d.center_pred:
        f0, f0d = W("fc_center.fc0.lin.weight"), W("fc_center.fc0.act.lin_dir.weight")
        d.off_c_fc0_t = blob.add(np.concatenate([f0, f0d @ f0], 0).T)
        misc = np.concatenate([W("fc_center.lin1.weight").reshape(-1), W("fc_center.shortcut.weight").reshape(-1),
                               W("fc_center.act2.lin_dir.weight").reshape(-1)])
        assert misc.size == h + cd + 1
        d.off_c_misc = blob.add(misc)
    else:
        d.off_c_fc0_t = blob.add(np.zeros((cd, 2 * h)))
        d.off_c_misc = blob.add(np.zeros(h + cd + 1))

    d.dec_num_linear = 0
    d.dec_latent_in = -1
    if dec_w is not None:
        lat, pe = dec_cfg["latent_size"], dec_cfg["pe_dim"]
        assert lat == cd and pe == cd + 1, "inner_deepsdf decoder: latent_size == c_dim and pe_dim == c_dim + 1"
        dims = list(dec_cfg["dims"])
        width = dims[0]
        assert all(x == width for x in dims), "uniform decoder width expected"
        nl = len(dims) + 1
        assert nl <= 12
        lin = list(dec_cfg["latent_in"])
        assert len(lin) <= 1
        li = lin[0] if lin else -1
        assert li != 0 and li != nl - 1 and (li < 0 or li >= 2)
        d.dec_num_linear, d.dec_width, d.dec_latent_in = nl, width, li
        u = lat + pe

        def DW(layer):
            if dec_cfg["weight_norm"] and layer in dec_cfg["norm_layers"]:
                g, v = _f64(dec_w[f"lin{layer}.weight_g"]), _f64(dec_w[f"lin{layer}.weight_v"])
                return g * v / np.linalg.norm(v, axis=1, keepdims=True)
            return _f64(dec_w[f"lin{layer}.weight"])

        def code_layer(layer, wu, bias):
            d.off_dec_inv_t[layer] = blob.add(wu[:, :lat].T)
            d.off_dec_so3_t[layer] = blob.add(wu[:, lat:2 * lat].T)
            d.off_dec_len[layer] = blob.add(wu[:, 2 * lat])
            d.off_dec_b[layer] = blob.add(bias)

        for layer in range(nl):
            w, b = DW(layer), _f64(dec_w[f"lin{layer}.bias"])
            if layer == 0:
                assert w.shape == (width, u)
                code_layer(0, w, b)
                d.off_dec_w[0] = 0
            elif layer == li:
                hprev = width - u
                kin = (hprev + 3) // 4 * 4
                assert w.shape == (width, width)
                main = np.zeros((width, kin))
                main[:, :hprev] = w[:, :hprev]
                d.off_dec_w[layer] = blob.add(main)
                code_layer(layer, w[:, hprev:], b)
            elif layer + 1 == li:
                hout = width - u
                outp = (hout + 3) // 4 * 4
                assert w.shape == (hout, width)
                wp, bp = np.zeros((outp, width)), np.zeros(outp)
                wp[:hout], bp[:hout] = w, b
                d.off_dec_w[layer], d.off_dec_b[layer] = blob.add(wp), blob.add(bp)
            else:
                d.off_dec_w[layer], d.off_dec_b[layer] = blob.add(w), blob.add(b)
    out = blob.finish()
    d.blob_floats = out.size
    return d, out
