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
    """matcher_new.py:85-98: mutual nearest neighbours.  desc [1,D,n] (the reference's transposed call convention).  Two launches: the cosine scores and
    ls_nn_match_f32 (both arg-maxes and both mutual checks)."""
    a, b = ops.nn_match(ops.cosine_scores(desc0[0].T.contiguous(), desc1[0].T.contiguous()))
    return {"matches0": a.squeeze(), "matches1": b.squeeze()}     # (the reference squeezes its [1, n] results: 0-dim for n == 1)


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters):
    """matcher_new.py:12-18 (kept for callers that want the transport plan itself; sinkhorn_matcher runs the loop inside one HIP launch)."""
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
    """matcher_new.py:45-71 (not selected by the evals, which use 'sequential').  Two launches: the cosine scores and ls_sinkhorn_match_f32 -- the coupling
    matrix in LDS, the 100 log-space iterations, the mutual arg-maxes and the threshold test in one workgroup-resident kernel (round 5: 200 dependent
    ATen logsumexp launches)."""
    a, b = ops.sinkhorn_match(ops.cosine_scores(desc0[0].T.contiguous(), desc1[0].T.contiguous()), desc_dim ** 0.5, alpha=1.0, iters=100,
                              match_threshold=match_threshold)
    return {"matches0": a.squeeze(), "matches1": b.squeeze()}
