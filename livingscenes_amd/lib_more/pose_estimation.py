"""Mirror of /root/reference/lib_more/pose_estimation.py for the registration path: kabsch_transformation_estimation
and transformation_residuals run in the HIP library (csrc/match.hip); the error metrics are a handful of 3x3 ops and
stay in torch.  Same names / argument meaning / return shapes as the reference (GUI helpers are out of scope)."""
import numpy as np  # noqa: F401  (the reference leaks np / torch through `from ... import *`, eval_3rscan.py:18)
import torch

from .. import ops
from ..lib_math.torch_se3 import inverse, transform


def kabsch_transformation_estimation(x1, x2, weights=None, normalize_w=True, eps=1e-7, best_k=0, w_threshold=0):
    """pose_estimation.py:29-102.  x1,x2 [b,n,3] (+ weights [b,n]) -> R [b,3,3], t [b,3,1], residuals [b,n], flag.
    The default call pattern (weights None or given, no best_k / w_threshold) is one kernel launch.  The non-default paths
    prepare the weights exactly as :49-66 do -- ones when None, divide by (sum + eps), best_k selection by batch item 0's
    weights, zeroing of NORMALISED weights below w_threshold without renormalising -- and hand them to the kernel as final.
    flag mirrors the reference: True iff its SVD branch would have failed (non-finite covariance); rank-deficient covariances
    return a valid least-squares rotation with flag False, like torch.svd."""
    if eps != 1e-7:
        raise NotImplementedError("eps != 1e-7 is not used by the reference's call sites")
    raw = (not normalize_w) or best_k > 0 or w_threshold > 0
    if raw:
        if weights is None:
            weights = torch.ones(x1.shape[0], x1.shape[1], dtype=x1.dtype, device=x1.device)   # :49-50
        if normalize_w:
            weights = weights / (weights.sum(1, keepdim=True) + eps)                            # :52-54
        if best_k > 0:   # :58-62 (np.argpartition of batch item 0; the SET of the best_k largest, order is irrelevant to Kabsch)
            idx = torch.topk(weights[0], best_k).indices.sort().values
            weights, x1, x2 = weights[:, idx], x1[:, idx], x2[:, idx]
        if w_threshold > 0:                                                                     # :64-65
            weights = torch.where(weights < w_threshold, torch.zeros_like(weights), weights)
    R, t, res, status = ops.kabsch(x1, x2, weights, return_flags=True, raw_weights=raw)
    return R, t, res, (status == 3).any()  # 0-dim bool tensor: truthy like the reference's flag, no host sync unless inspected


def transformation_residuals(x1, x2, R, t):
    """pose_estimation.py:105-121."""
    return torch.norm((R @ x1.transpose(1, 2) + t).transpose(1, 2) - x2, dim=2)


def solve_R(f1, f2):
    """pose_estimation.py:11-27: un-weighted, un-centred rotation fit (f [(b,) m, 3]): H = f1^T f2 = U S V^T, R = V diag(1, 1, det(V U^T)) U^T.
    Runs on the device's batched Kabsch kernel (the reference's torch.svd would be a host round trip here): the point sets are mirrored
    (x, -x), which makes both centroids zero, so the kernel's centred covariance IS 2 f1^T f2 / (2m) -- a positive multiple of H, same rotation."""
    squeeze = f1.dim() == 2
    if squeeze:
        f1, f2 = f1[None], f2[None]
    R, _, _ = ops.kabsch(torch.cat([f1, -f1], 1), torch.cat([f2, -f2], 1))
    return R[0] if squeeze else R


def inverse_3d_transform(tsfm):
    """pose_estimation.py:123-138 (4x4)."""
    out = torch.zeros_like(tsfm)
    Rt = tsfm[:, :3, :3].transpose(-1, -2)
    out[:, :3, :3], out[:, :3, 3:4], out[:, 3, 3] = Rt, -(Rt @ tsfm[:, :3, 3:4]), 1
    return out


def solve_transform_from_latent(code1, code2):
    """pose_estimation.py:140-154."""
    R = solve_R(code1["z_so3"], code2["z_so3"])
    t = code2["t"] - torch.einsum("bnm,bjm->bjn", R, code1["t"])
    T = torch.eye(4, device=R.device).unsqueeze(0).repeat(R.shape[0], 1, 1)
    T[:, :3, :3], T[:, :3, 3:4] = R, t.transpose(-1, -2)
    return T


def rotation_error(R1, R2):
    """pose_estimation.py:157-180: geodesic angle in degrees, [b,1]."""
    tr = torch.diagonal(R1.transpose(1, 2) @ R2, dim1=-2, dim2=-1).sum(-1)
    return 180.0 * torch.acos(torch.clamp(((tr - 1) / 2).unsqueeze(1), -1, 1)) / torch.pi


def translation_error(t1, t2):
    """pose_estimation.py:183-196."""
    return torch.norm(t1 - t2, dim=(-2, -1))


def evaluate_transform(gt_tsfm, pred_tsfm):
    """pose_estimation.py:199-211."""
    return (rotation_error(gt_tsfm[:, :3, :3], pred_tsfm[:, :3, :3]),
            translation_error(gt_tsfm[:, :3, 3], pred_tsfm[:, :3, 3]))


def compute_transformation_error(pc1, pc2, pred_tsfm, gt_tsfm, thres=0.2):
    """pose_estimation.py:214-233: endpoint RMSE in both directions."""
    e12 = transform(pred_tsfm, pc1) - transform(gt_tsfm, pc1)
    e21 = transform(inverse(pred_tsfm), pc2) - transform(inverse(gt_tsfm), pc2)
    return (torch.cat([e12, e21], dim=1) ** 2).mean().sqrt()


def huber_norm_weights(x, b=0.02):
    """pose_estimation.py:258-272."""
    r = torch.where(x <= b, x ** 2, 2 * b * x - b ** 2)
    x = torch.where(x == 0, torch.ones_like(x), x)
    return torch.sqrt(r) / x


def get_robust_res(res, b):
    """pose_estimation.py:274-289."""
    res = res.view(-1, 1, 1)
    w = huber_norm_weights(torch.abs(res), b=b)
    return w * res, w ** 2
