"""Mirror of /root/reference/lib_more/matcher_new.py: the five matchers with the reference's signatures, returning
{'matches0', 'matches1'} (int64, -1 = unmatched).  Score matrices, the greedy assignment loop and the Kabsch residual
matrices run in the HIP library (csrc/match.hip) -- one launch each, no host round trip per iteration."""
import torch
import torch.nn.functional as F

from .. import ops


def sequential_matcher(m0, m1):
    """matcher_new.py:109-139: cosine scores of the invariant codes + greedy global-max assignment."""
    a, b = ops.greedy_match(ops.cosine_scores(m0, m1))
    return {"matches0": a, "matches1": b}


def _residuals(src_codes, tgt_codes):
    return ops.kabsch_residual_matrix(src_codes["z_so3"].detach(), tgt_codes["z_so3"].detach())


def sim3_seq_matcher(src_codes, tgt_codes):
    """matcher_new.py:142-184: scores = cosine / (mean Kabsch residual + 1e-5)."""
    sim = ops.cosine_scores(src_codes["z_inv"].detach(), tgt_codes["z_inv"].detach())
    a, b = ops.greedy_match(sim / (_residuals(src_codes, tgt_codes) + 1e-5))
    return {"matches0": a, "matches1": b}


def eq_seq_matcher(src_codes, tgt_codes):
    """matcher_new.py:188-230: scores = 1 / (mean Kabsch residual + 1e-5)."""
    a, b = ops.greedy_match(1 / (_residuals(src_codes, tgt_codes) + 1e-5))
    return {"matches0": a, "matches1": b}


def mutual_check(m0, m1):
    """matcher_new.py:100-105."""
    inds0 = torch.arange(m0.shape[-1], device=m0.device)
    loop = torch.gather(m1, -1, torch.where(m0 > -1, m0, m0.new_tensor(0)))
    return torch.where((m0 > -1) & (inds0 == loop), m0, m0.new_tensor(-1))


def nn_matcher(desc0, desc1):
    """matcher_new.py:85-98: mutual nearest neighbours.  desc [1,D,n] (the reference's transposed call convention)."""
    sim = ops.cosine_scores(desc0[0].T.contiguous(), desc1[0].T.contiguous())[None]
    m0 = sim.argmax(dim=2)
    m1 = sim.argmax(dim=1)
    m0 = mutual_check(m0, m1)
    m1 = mutual_check(m1, m0)
    return {"matches0": m0.squeeze(), "matches1": m1.squeeze()}


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters):
    """matcher_new.py:20-40 (SuperGlue's log-space optimal transport with a dustbin row/column)."""
    b, m, n = scores.shape
    ms, ns = scores.new_tensor(float(m)), scores.new_tensor(float(n))
    top = torch.cat([scores, alpha.expand(b, m, 1)], -1)
    bot = torch.cat([alpha.expand(b, 1, n), alpha.expand(b, 1, 1)], -1)
    Zc = torch.cat([top, bot], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    return log_sinkhorn_iterations(Zc, log_mu, log_nu, iters) - norm


def sinkhorn_matcher(desc0, desc1, desc_dim=256, match_threshold=0.0):
    """matcher_new.py:45-71: not selected by the evals (they use 'sequential'); small dense torch ops on the device."""
    scores = ops.cosine_scores(desc0[0].T.contiguous(), desc1[0].T.contiguous())[None] / desc_dim ** 0.5
    Z = log_optimal_transport(scores, torch.tensor(1.0, device=scores.device), iters=100)
    max0, max1 = Z[:, :-1, :-1].max(2), Z[:, :-1, :-1].max(1)
    i0, i1 = max0.indices, max1.indices
    ar0 = torch.arange(i0.shape[1], device=i0.device)[None]
    ar1 = torch.arange(i1.shape[1], device=i1.device)[None]
    mutual0 = ar0 == i1.gather(1, i0)
    mutual1 = ar1 == i0.gather(1, i1)
    zero = Z.new_tensor(0)
    ms0 = torch.where(mutual0, max0.values.exp(), zero)
    valid0 = mutual0 & (ms0 > match_threshold)
    valid1 = mutual1 & valid0.gather(1, i1)
    return {"matches0": torch.where(valid0, i0, i0.new_tensor(-1)).squeeze(),
            "matches1": torch.where(valid1, i1, i1.new_tensor(-1)).squeeze()}
