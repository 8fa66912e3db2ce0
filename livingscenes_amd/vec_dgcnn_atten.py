"""VecDGCNN_att -- the reference's encoder class (same constructor arguments, same state_dict keys, same
forward signature) whose forward runs entirely in the HIP library.

Mirrors /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:22-252 (class VecDGCNN_att) and the
parameter containers of vec_layers.py (VecLinear :34, VecActivation :214, VecLinearNormalizeActivate :488,
VecResBlock :537).  Only parameters live here; there is no PyTorch compute path (no CPU fallback).
"""
import math

import torch
from torch import nn

from . import ops, packing


# Every nn.Parameter / sub-module registration in the process bumps this counter (torch's global registration hooks): hip_model()
# re-walks its parameter list when it moved (setattr of a new Parameter, a swapped sub-module, weight-norm re-wrapping ...).
_REGISTRATIONS = [0]


def _bump(*_args):
    _REGISTRATIONS[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump)
torch.nn.modules.module.register_module_module_registration_hook(_bump)


class VecLinear(nn.Module):
    """Parameter container of VecLinear (so3 mode, vector path only): weight [v_out, v_in], no bias."""

    def __init__(self, v_in, v_out, mode="so3"):
        super().__init__()
        assert mode.lower() == "so3", "only so3 layers are on the hot path"
        self.v_in, self.v_out = v_in, v_out
        self.weight = nn.Parameter(torch.empty(v_out, v_in))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # vec_layers.py:114-119


class VecActivation(nn.Module):
    def __init__(self, in_features, act_func=None, shared_nonlinearity=False, mode="so3"):
        super().__init__()
        self.lin_dir = VecLinear(in_features, 1 if shared_nonlinearity else in_features, mode=mode)


class VecLinearNormalizeActivate(nn.Module):
    def __init__(self, in_features, out_features, act_func=None, shared_nonlinearity=False, mode="so3"):
        super().__init__()
        self.lin = VecLinear(in_features, out_features, mode=mode)
        self.act = VecActivation(out_features, act_func, shared_nonlinearity, mode)


VecLNA = VecLinearNormalizeActivate


class VecResBlock(nn.Module):
    def __init__(self, in_features, out_features, hidden_features, act_func=None, mode="so3"):
        super().__init__()
        assert in_features != out_features
        self.fc0 = VecLNA(in_features, hidden_features, act_func, mode=mode)
        self.lin1 = VecLinear(hidden_features, out_features, mode=mode)
        self.act2 = VecActivation(out_features, act_func, mode=mode)
        self.shortcut = VecLinear(in_features, out_features, mode=mode)


class VecDGCNN_att(nn.Module):
    def __init__(self, c_dim=256, num_layers=8, feat_dim=(32, 32, 64, 64, 128, 256, 512, 512),
                 down_sample_layers=(2, 4, 6), down_sample_factor=(4, 4, 4), atten_start_layer=2,
                 atten_multi_head_c=16, use_res_global_conv=True, res_global_start_layer=2, num_knn=16,
                 num_knn_early=-1, knn_early_layers=-1, scale_factor=640.0, leak_neg_slope=0.2, use_dg=True,
                 center_pred=False, center_pred_scale=False, z_so3_as_Omtx=False):
        super().__init__()
        assert use_dg, "only the dynamic-graph encoder is implemented on the HIP path"
        assert not z_so3_as_Omtx, "z_so3_as_Omtx is not used by the released config"
        assert num_knn_early < 0 or num_knn_early == num_knn
        assert len(down_sample_factor) == len(down_sample_layers) and len(feat_dim) == num_layers
        assert atten_start_layer >= 1, "first layers should use naive DGCNN"
        feat_dim = list(feat_dim)
        self.cfg = dict(c_dim=c_dim, num_layers=num_layers, feat_dim=feat_dim,
                        down_sample_layers=list(down_sample_layers), down_sample_factor=list(down_sample_factor),
                        atten_start_layer=atten_start_layer, atten_multi_head_c=atten_multi_head_c,
                        use_res_global_conv=use_res_global_conv, res_global_start_layer=res_global_start_layer,
                        num_knn=num_knn, scale_factor=scale_factor, leak_neg_slope=leak_neg_slope, use_dg=use_dg,
                        center_pred=center_pred, center_pred_scale=center_pred_scale)
        self.c_dim, self.k, self.num_layers, self.feat_dim = c_dim, num_knn, num_layers, feat_dim
        self.scale_factor, self.center_pred, self.center_pred_scale = scale_factor, center_pred, center_pred_scale
        self.global_conv_list, self.V_list = nn.ModuleList(), nn.ModuleList()
        self.Q_list, self.K_list = nn.ModuleList(), nn.ModuleList()
        for i in range(num_layers):
            self.V_list.append(VecLNA(3 if i == 0 else feat_dim[i - 1] * 2, feat_dim[i]))
            if use_res_global_conv and i >= res_global_start_layer:
                self.global_conv_list.append(VecLNA(feat_dim[i] * 2, feat_dim[i]))
            if i >= atten_start_layer:
                assert feat_dim[i] % atten_multi_head_c == 0
                self.Q_list.append(VecLNA(feat_dim[i - 1], feat_dim[i]))
                self.K_list.append(VecLNA(feat_dim[i - 1] * 2, feat_dim[i]))
            else:
                self.Q_list.append(None), self.K_list.append(None)
        self.conv_c = VecLNA(feat_dim[-1], c_dim, shared_nonlinearity=True)
        self.fc_inv = VecLinear(c_dim, c_dim)
        if center_pred:
            self.fc_center = VecResBlock(c_dim, 1, c_dim // 2)
        self._hip = None
        self._hip_key = None

    # ------------------------------------------------------------------ HIP model cache
    def hip_model(self, decoder=None):
        """Packed device model (encoder [+ decoder]); rebuilt when any parameter tensor changed (storage or in-place version).
        The parameter LIST is cached per decoder object (walking the module tree costs more than the check itself: this runs on
        every encode), the per-tensor (data_ptr, _version) check is not."""
        cache = getattr(self, "_hip_plist", None)
        if cache is None or cache[0] is not decoder or cache[2] != _REGISTRATIONS[0]:
            # (re-)walk the trees: first use, another decoder, or some module registered a parameter / sub-module since the last walk
            # (a replaced nn.Parameter or a swapped sub-module would otherwise keep matching the stale list)
            plist = list(self.parameters()) + ([] if decoder is None else list(decoder.parameters()))
            self._hip_plist = cache = (decoder, plist, _REGISTRATIONS[0])
        plist = cache[1]
        key = tuple([(p.data_ptr(), p._version) for p in plist])
        if self._hip is None or self._hip_key != key:
            if self._hip is not None:
                self._hip.close()
            dev = plist[0].device
            enc_w = {k: v for k, v in self.state_dict().items()}
            dec_w = dec_cfg = None
            if decoder is not None:
                dec_w, dec_cfg = decoder.state_dict(), decoder.cfg
            desc, blob = packing.pack_model(enc_w, self.cfg, dec_w, dec_cfg)
            self._hip = ops.HipModel(desc, blob, dev)
            self._hip_key = key
        return self._hip

    def forward(self, x):
        """x [B,3,N] -> (center [B,1,3], scale [B], z_so3 [B,c_dim,3], z_inv [B,c_dim])   (vec_dgcnn_atten.py:177-252);
        without center_pred the tuple is (scale, z_so3, z_inv) as in the reference."""
        z_so3, z_inv, s, t = self.hip_model().encode(x, pre_normalised=True)
        if self.center_pred:
            return t.unsqueeze(1), s, z_so3, z_inv
        return s, z_so3, z_inv
