"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU restatement of the matcher / registration operators.

Algorithms follow /root/reference/lib_more/matcher_new.py, lib_more/pose_estimation.py and
lib_math/torch_se3.py (line numbers cited per function); pinned against the imported reference
functions by tests/golden/make_golden.py -> tests/golden/*.npz.  ICP restates pytorch3d 0.7.4's
``iterative_closest_point`` (un-vendored; PARITY UNPINNED) as called at
lib_more/more_solver.py:182-187.  Only tests/, smoke() and bench.py's cpu_baseline leg import this.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- greedy matchers
def _greedy_assign(score, n_src, n_tgt):
    """The shared greedy loop of matcher_new.py:121-136 (and :166-181, :212-227): renormalise by
    (max + 1e-5), take the FIRST row-major arg-max, record it, delete its row and column."""
    rows = list(range(n_src))
    cols = list(range(n_tgt))
    m0 = -torch.ones(n_src, dtype=torch.long)
    m1 = -torch.ones(n_tgt, dtype=torch.long)
    S = score.clone()
    for _ in range(min(n_src, n_tgt)):
        S = S / (S.max() + 1e-5)
        hit = (S == S.max()).nonzero()
        r, c = int(hit[0, 0]), int(hit[0, 1])
        m0[rows[r]] = cols[c]
        m1[cols[c]] = rows[r]
        del rows[r], cols[c]
        keep_r = [i for i in range(S.shape[0]) if i != r]
        keep_c = [j for j in range(S.shape[1]) if j != c]
        S = S[keep_r][:, keep_c]
    return {"matches0": m0, "matches1": m1}


def sequential_matcher(m0, m1):
    """matcher_new.py:109-139: cosine scores of L2-normalised invariant codes + greedy loop."""
    a = F.normalize(m0, p=2, dim=1)
    b = F.normalize(m1, p=2, dim=1)
    return _greedy_assign(a @ b.T, a.shape[0], b.shape[0])


def cosine_scores(m0, m1):
    return F.normalize(m0, p=2, dim=1) @ F.normalize(m1, p=2, dim=1).T


def nn_matcher(desc0, desc1):
    """matcher_new.py:85-107 (find_nn with no thresholds + mutual_check). desc [1,D,n]."""
    a = F.normalize(desc0, p=2, dim=1)
    b = F.normalize(desc1, p=2, dim=1)
    sim = torch.einsum("bdn,bdm->bnm", a, b)
    i0 = sim.argmax(dim=2)
    i1 = sim.argmax(dim=1)
    ar0 = torch.arange(i0.shape[-1])[None]
    ar1 = torch.arange(i1.shape[-1])[None]
    ok0 = i1.gather(1, i0) == ar0
    out0 = torch.where(ok0, i0, torch.full_like(i0, -1))
    # second mutual check runs against the already filtered matches0 (matcher_new.py:93-94)
    loop = out0.gather(1, i1)
    ok1 = loop == ar1
    out1 = torch.where(ok1, i1, torch.full_like(i1, -1))
    return {"matches0": out0.squeeze(), "matches1": out1.squeeze()}


def log_optimal_transport(scores, alpha, iters):
    """matcher_new.py:12-40: couplings = [[scores, alpha], [alpha, alpha]] ([b, m+1, n+1]), log_mu / log_nu with the dustbin masses, `iters` iterations of
    u = log_mu - logsumexp(Z + v, dim=2); v = log_nu - logsumexp(Z + u, dim=1); returns Z + u + v - norm."""
    b, m, n = scores.shape
    ms, ns = float(m), float(n)
    Z = torch.full((b, m + 1, n + 1), float(alpha), dtype=scores.dtype)
    Z[:, :m, :n] = scores
    norm = -torch.tensor(ms + ns, dtype=scores.dtype).log()
    log_mu = torch.cat([norm.expand(m), torch.tensor(ns, dtype=scores.dtype).log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), torch.tensor(ms, dtype=scores.dtype).log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1) - norm


def sinkhorn_matcher(desc0, desc1, desc_dim=256, match_threshold=0.0, iters=100):
    """matcher_new.py:45-71. desc [1,D,n]: normalised descriptors, scores / sqrt(desc_dim), optimal transport with alpha = 1, mutual arg-maxes of the
    inner block, exp(max0) > match_threshold."""
    a = F.normalize(desc0, p=2, dim=1)
    b = F.normalize(desc1, p=2, dim=1)
    scores = torch.einsum("bdn,bdm->bnm", a, b) / desc_dim ** .5
    Z = log_optimal_transport(scores, 1.0, iters)
    max0, max1 = Z[:, :-1, :-1].max(2), Z[:, :-1, :-1].max(1)
    i0, i1 = max0.indices, max1.indices
    mutual0 = torch.arange(i0.shape[1])[None] == i1.gather(1, i0)
    mutual1 = torch.arange(i1.shape[1])[None] == i0.gather(1, i1)
    ms0 = torch.where(mutual0, max0.values.exp(), torch.zeros(()))
    valid0 = mutual0 & (ms0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, i1)
    return {"matches0": torch.where(valid0, i0, torch.full_like(i0, -1)).squeeze(),
            "matches1": torch.where(valid1, i1, torch.full_like(i1, -1)).squeeze(), "Z": Z[0]}


def kabsch_residual_matrix(src_so3, tgt_so3):
    """res_mat of matcher_new.py:150-156 / :196-202: mean Kabsch residual between every pair."""
    n, m = src_so3.shape[0], tgt_so3.shape[0]
    res = torch.zeros(n, m)
    for i in range(n):
        _, _, r, _ = kabsch_transformation_estimation(src_so3[i][None].repeat_interleave(m, dim=0), tgt_so3)
        res[i] = r.mean(dim=1)
    return res


def eq_seq_matcher(src_codes, tgt_codes):
    """matcher_new.py:188-230: score = 1 / (res + 1e-5), then the greedy loop."""
    res = kabsch_residual_matrix(src_codes["z_so3"], tgt_codes["z_so3"])
    return _greedy_assign(1 / (res + 1e-5), res.shape[0], res.shape[1])


def sim3_seq_matcher(src_codes, tgt_codes):
    """matcher_new.py:142-184: score = cosine / (res + 1e-5), then the greedy loop."""
    sim = cosine_scores(src_codes["z_inv"], tgt_codes["z_inv"])
    res = kabsch_residual_matrix(src_codes["z_so3"], tgt_codes["z_so3"])
    return _greedy_assign(sim / (res + 1e-5), res.shape[0], res.shape[1])


# ----------------------------------------------------------------------------- Kabsch
def transformation_residuals(x1, x2, R, t):
    """pose_estimation.py:105-121."""
    rec = torch.matmul(R, x1.transpose(1, 2)) + t
    return torch.norm(rec.transpose(1, 2) - x2, dim=2)


def kabsch_transformation_estimation(x1, x2, weights=None, normalize_w=True, eps=1e-7):
    """pose_estimation.py:29-102 (best_k = 0, w_threshold = 0): weighted Kabsch, batched.
    Returns R [b,3,3], t [b,3,1], residuals [b,n], flag."""
    b, n, _ = x1.shape
    if weights is None:
        weights = torch.ones(b, n, dtype=x1.dtype)
    if normalize_w:
        weights = weights / (weights.sum(dim=1, keepdim=True) + eps)
    wcol = weights.unsqueeze(2)
    denom = wcol.sum(dim=1).unsqueeze(1) + eps
    mu1 = torch.matmul(wcol.transpose(1, 2), x1) / denom
    mu2 = torch.matmul(wcol.transpose(1, 2), x2) / denom
    c1, c2 = x1 - mu1, x2 - mu2
    cov = torch.matmul(c1.transpose(1, 2), torch.matmul(torch.diag_embed(wcol.squeeze(2)), c2))
    u, s, v = torch.svd(cov)
    det = torch.det(torch.matmul(v.transpose(1, 2), u.transpose(1, 2)))
    D = torch.diag_embed(torch.cat((torch.ones(b, 2, dtype=x1.dtype), det.unsqueeze(1)), 1))
    R = torch.matmul(v, torch.matmul(D, u.transpose(1, 2)))
    t = mu2.transpose(1, 2) - torch.matmul(R, mu1.transpose(1, 2))
    return R, t, transformation_residuals(x1, x2, R, t), False


# ----------------------------------------------------------------------------- SE(3) + metrics
def se3_inverse(g):
    """lib_math/torch_se3.py:10-25."""
    Rt = g[..., 0:3, 0:3].transpose(-1, -2)
    return torch.cat([Rt, Rt @ -g[..., 0:3, 3][..., None]], dim=-1)


def se3_concatenate(a, b):
    """lib_math/torch_se3.py:28-49."""
    R = a[..., :3, :3] @ b[..., :3, :3]
    t = a[..., :3, :3] @ b[..., :3, 3][..., None] + a[..., :3, 3][..., None]
    return torch.cat([R, t], dim=-1)


def se3_transform(g, pts):
    """lib_math/torch_se3.py:52-78 (points only)."""
    return torch.matmul(pts, g[..., :3, :3].transpose(-1, -2)) + g[..., :3, 3][..., None, :]


def Rt_to_SE3(R, t):
    """lib_math/torch_se3.py:81-92."""
    out = torch.zeros(R.shape[0], 4, 4)
    out[:, 3, 3] = 1
    out[:, :3, :3] = R
    out[:, :3, 3:4] = t
    return out


def rotation_error(R1, R2):
    """pose_estimation.py:157-180 (degrees, [b,1])."""
    tr = torch.einsum("bii->b", torch.matmul(R1.transpose(1, 2), R2))
    e = torch.clamp(((tr - 1) / 2).unsqueeze(1), -1, 1)
    return 180.0 * torch.acos(e) / torch.pi


def translation_error(t1, t2):
    """pose_estimation.py:183-196."""
    return torch.norm(t1 - t2, dim=(-2, -1))


def compute_transformation_error(pc1, pc2, pred, gt):
    """pose_estimation.py:214-233: endpoint RMSE, both directions."""
    e12 = se3_transform(pred, pc1) - se3_transform(gt, pc1)
    e21 = se3_transform(se3_inverse(pred), pc2) - se3_transform(se3_inverse(gt), pc2)
    return (torch.cat([e12, e21], dim=1) ** 2).mean().sqrt()


# ----------------------------------------------------------------------------- ICP (pytorch3d, un-vendored)
def _nn1(X, Y):
    """1-NN of each X row in Y under the canonical squared distance (contract=0), first minimum."""
    from . import canon
    B, N, _ = X.shape
    xr = X.numpy().reshape(B, N, 3, 1)
    yr = Y.numpy().reshape(B, Y.shape[1], 3, 1)
    idx = canon.knn_c(xr, yr, 1, contract=0)[..., 0]
    return torch.from_numpy(idx.astype(np.int64))


def corresponding_points_alignment(X, Y):
    """pytorch3d.ops.corresponding_points_alignment (rigid, uniform weights, no reflection):
    row-vector convention Y ~ X R + T."""
    mx, my = X.mean(1, keepdim=True), Y.mean(1, keepdim=True)
    Xc, Yc = X - mx, Y - my
    cov = torch.bmm(Xc.transpose(1, 2), Yc) / X.shape[1]
    U, S, V = torch.svd(cov)
    E = torch.eye(3)[None].repeat(X.shape[0], 1, 1)
    E[:, 2, 2] = torch.det(torch.bmm(U, V.transpose(1, 2)))
    R = torch.bmm(torch.bmm(U, E), V.transpose(1, 2))
    T = my[:, 0] - torch.bmm(mx, R)[:, 0]
    return R, T


def iterative_closest_point(X, Y, R0, T0, max_iterations=100, relative_rmse_thr=1e-6):
    """pytorch3d.ops.iterative_closest_point(X, Y, init_transform=SimilarityTransform(R0,T0,1)) with
    default arguments, as called at lib_more/more_solver.py:182-184.  Row-vector convention
    Xt = X R + T.  Returns (R, T, rmse, iterations, converged)."""
    R, T = R0.clone(), T0.clone()
    Xt = torch.bmm(X, R) + T[:, None]
    prev = None
    converged = False
    it = 0
    rmse = torch.zeros(X.shape[0])
    for it in range(max_iterations):
        nn = _nn1(Xt, Y)
        Yn = torch.gather(Y, 1, nn[..., None].expand(-1, -1, 3))
        R, T = corresponding_points_alignment(X, Yn)
        Xt = torch.bmm(X, R) + T[:, None]
        rmse = ((Xt - Yn) ** 2).sum(2).mean(1).sqrt()
        if prev is not None:
            rel = (prev - rmse) / prev
            if bool((rel <= relative_rmse_thr).all()):
                converged = True
                break
        prev = rmse
    return R, T, rmse, it + 1, converged
