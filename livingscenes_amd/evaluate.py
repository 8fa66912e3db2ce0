"""Evaluation metrics of the reference's evaluate.py that the 3RScan relocalisation loop uses.

chamfer_distance_torch (/root/reference/evaluate.py:111-123): both clouds are compared after the PREDICTED transform -- the
source moved by the prediction against the target, and the target against itself moved by prediction o inverse(ground truth) --
as mean nearest-neighbour SQUARED distance in each direction, summed.  The reference materialises the [n, m] squared-distance
matrix; here the nearest neighbours come from the library's raw-cloud k-NN (ls_knn_f32, K = 1: wave-per-query kernel), whose
distance is the same (dx^2 + dy^2 + dz^2) chain.  HIP tensors only, like every operator of this package.
"""
import torch

from . import ops
from .lib_math import torch_se3


def _nn_sq_dist(a, b):
    """a [B,n,3], b [B,m,3] -> [B,n] squared distance of every a-point to its nearest b-point"""
    _, d = ops.knn(a.float().contiguous().unsqueeze(-1), b.float().contiguous().unsqueeze(-1), 1, return_dist=True)
    return d[..., 0]


def chamfer_distance_torch(src, ref, pred_tsfm, gt_tsfm):
    src_transformed = torch_se3.transform(pred_tsfm, src)
    ref_inv_transformed = torch_se3.transform(torch_se3.concatenate(pred_tsfm, torch_se3.inverse(gt_tsfm)), ref)
    dist_src = _nn_sq_dist(src_transformed, ref)
    dist_ref = _nn_sq_dist(ref, ref_inv_transformed)
    return dist_src.mean(dim=1) + dist_ref.mean(dim=1)
