"""DeepSDF_Decoder -- parameter container with the reference's constructor and state_dict keys
(/root/reference/lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py:9-76: lin{l}.weight_g / weight_v / bias for
the weight-normed layers, lin{l}.weight / bias otherwise).  The forward lives in the HIP library (csrc/sdf.hip +
csrc/gemm.hip) and is reached through model_utils.FieldWrapper; there is no PyTorch compute path."""
import math

import torch
from torch import nn


class _WNLinear(nn.Module):
    """state_dict-compatible with nn.utils.weight_norm(nn.Linear(i, o)) (old-style hook: weight_g [o,1], weight_v [o,i])."""

    def __init__(self, i, o):
        super().__init__()
        v = torch.empty(o, i)
        nn.init.kaiming_uniform_(v, a=math.sqrt(5))
        self.bias = nn.Parameter(torch.empty(o).uniform_(-1 / math.sqrt(i), 1 / math.sqrt(i)))
        self.weight_g = nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(v)


class DeepSDF_Decoder(nn.Module):
    def __init__(self, latent_size, dims, dropout=None, dropout_prob=0.0, norm_layers=(), latent_in=(), weight_norm=False,
                 xyz_in_all=None, use_tanh=False, latent_dropout=False, pe_dim=3):
        super().__init__()
        assert not xyz_in_all and not use_tanh and not latent_dropout, "not used by the released config"
        self.cfg = dict(latent_size=latent_size, dims=list(dims), norm_layers=list(norm_layers or []),
                        latent_in=list(latent_in or []), weight_norm=weight_norm, pe_dim=pe_dim)
        d = [latent_size + pe_dim] + list(dims) + [1]
        self.num_layers = len(d)
        self.latent_size, self.pe_dim, self.latent_in, self.norm_layers = latent_size, pe_dim, list(latent_in or []), norm_layers
        for layer in range(self.num_layers - 1):
            out_dim = d[layer + 1] - d[0] if (layer + 1) in self.latent_in else d[layer + 1]
            if weight_norm and layer in (norm_layers or []):
                setattr(self, f"lin{layer}", _WNLinear(d[layer], out_dim))
            else:
                assert not (norm_layers and layer in norm_layers), "LayerNorm variant is not on the released path"
                setattr(self, f"lin{layer}", nn.Linear(d[layer], out_dim))

    def forward(self, inp, phase="val"):
        raise NotImplementedError("DeepSDF_Decoder runs inside the HIP library: call it through FieldWrapper "
                                  "(livingscenes_amd.model_utils), which folds the code into the first/skip layers")
