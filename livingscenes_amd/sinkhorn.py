"""Debiased Sinkhorn divergence between two point clouds, as the optimisation-based registration uses it:
``geomloss.SamplesLoss(loss='sinkhorn', p=2)`` at /root/reference/lib_more/more_solver.py:146,158.

geomloss is neither vendored nor installed here, so this module RESTATES its published algorithm from memory (defaults of
SamplesLoss('sinkhorn', p=2): cost |x-y|^2/2, blur 0.05, scaling 0.5, debias, uniform weights, epsilon-scaling from the squared
bounding-box diameter down to blur^2, symmetric (averaged) updates, one last extrapolation step that carries the gradient) --
**parity UNPINNED**; oracle/sinkhorn.py is the same definition on torch-CPU with autograd and only checks the kernels.
Every softmin runs in csrc/sinkhorn.hip (ls_sinkhorn_softmin_f32); the loop below only sequences ~45 launches.
"""
import math

import numpy as np
import torch

from ._lib import call, check, load, ptr, stream_ptr


def softmin(eps, x, y, h, need_grad=False):
    """-eps * logsumexp_j(h_j - |x_i - y_j|^2 / (2 eps)) for every row of x [N,3] against y [M,3]; optionally d/dx_i."""
    N, M = x.shape[0], y.shape[0]
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    grad = torch.empty(N, 3, dtype=torch.float32, device=x.device) if need_grad else None
    call(x.device, "ls_sinkhorn_softmin_f32", ptr(x), ptr(y), ptr(h), N, M, float(eps), ptr(out), ptr(grad), stream_ptr(x.device))
    return (out, grad) if need_grad else out


def epsilon_schedule(diameter, blur=0.05, scaling=0.5, p=2):
    return ([diameter ** p] + [float(np.exp(e)) for e in np.arange(p * math.log(diameter), p * math.log(blur), p * math.log(scaling))]
            + [blur ** p])


def _divergence(x, y, blur, scaling):
    """-> (loss float32 scalar tensor, d loss / d x [N,3]) for x [N,3], y [M,3] (contiguous fp32 on the GPU)."""
    N, M = x.shape[0], y.shape[0]
    both = torch.cat([x, y], 0)
    diameter = float((both.max(0)[0] - both.min(0)[0]).norm())
    eps_list = epsilon_schedule(max(diameter, 1e-6), blur, scaling)
    a_log = torch.full((N,), -math.log(N), dtype=torch.float32, device=x.device)
    b_log = torch.full((M,), -math.log(M), dtype=torch.float32, device=x.device)
    eps = eps_list[0]
    g_ab, f_ba = softmin(eps, y, x, a_log), softmin(eps, x, y, b_log)
    f_aa, g_bb = softmin(eps, x, x, a_log), softmin(eps, y, y, b_log)
    for eps in eps_list:
        ft_ba = softmin(eps, x, y, b_log + g_ab / eps)
        gt_ab = softmin(eps, y, x, a_log + f_ba / eps)
        ft_aa = softmin(eps, x, x, a_log + f_aa / eps)
        gt_bb = softmin(eps, y, y, b_log + g_bb / eps)
        f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
        f_aa, g_bb = 0.5 * (f_aa + ft_aa), 0.5 * (g_bb + gt_bb)
    # last extrapolation: the only step the gradient flows through (potentials on the right-hand side are constants)
    f_ba_l, d_ba = softmin(eps, x, y, b_log + g_ab / eps, need_grad=True)
    g_ab_l = softmin(eps, y, x, a_log + f_ba / eps)
    f_aa_l, d_aa = softmin(eps, x, x, a_log + f_aa / eps, need_grad=True)
    g_bb_l = softmin(eps, y, y, b_log + g_bb / eps)
    loss = (f_ba_l - f_aa_l).mean() + (g_ab_l - g_bb_l).mean()
    return loss, (d_ba - d_aa) / N


def _softmin_b(x, y, pot_y, logw, eps, prev=None, average=False, need_grad=False):
    """Batched softmin (csrc/optim.hip): x [P,N,3], y [P,M,3], pot_y [P,M] or None, eps [P] (<= 0: pair inactive -> prev)."""
    P, N, _ = x.shape
    M = y.shape[1]
    out = torch.empty(P, N, dtype=torch.float32, device=x.device)
    grad = torch.empty(P, N, 3, dtype=torch.float32, device=x.device) if need_grad else None
    call(x.device, "ls_sinkhorn_softmin_batched_f32", ptr(x), ptr(y), ptr(pot_y), float(logw), ptr(eps), ptr(prev), int(average), P, N, M,
         ptr(out), ptr(grad), stream_ptr(x.device))
    return (out, grad) if need_grad else out


def _softmin_multi(problems, eps, average):
    """Up to four independent batched softmins in ONE launch: problems = [(x, y, pot_y | None, logw, prev | None), ...] -> [out, ...]."""
    from ._lib import SoftminProblem
    import ctypes
    P = problems[0][0].shape[0]
    arr = (SoftminProblem * len(problems))()
    outs = []
    for i, (x, y, pot, logw, prev) in enumerate(problems):
        out = torch.empty(P, x.shape[1], dtype=torch.float32, device=x.device)
        outs.append(out)
        arr[i] = SoftminProblem(ptr(x), ptr(y), ptr(pot), ptr(prev), ptr(out), float(logw), x.shape[1], y.shape[1])
    call(problems[0][0].device, "ls_sinkhorn_softmin_multi_f32", ctypes.addressof(arr), len(problems), ptr(eps), int(average), P,
         stream_ptr(problems[0][0].device))
    return outs


def divergence_batch(x, y, blur=0.05, scaling=0.5, lmax=None, return_need=False):
    """Debiased Sinkhorn divergence of P cloud pairs in lock-step: x [P,N,3] (moving), y [P,M,3] -> (loss [P], d loss / d x [P,N,3]).
    Same definition as _divergence pair by pair: every pair follows ITS OWN epsilon schedule (it depends on the pair's bounding-box
    diameter, so the schedules differ in length); a pair whose schedule has ended is passed through unchanged by the kernel
    (eps <= 0).  The loop length is the longest schedule of the batch: read back from the device when ``lmax`` is None (one host
    sync), or given by the caller (e.g. the value of an earlier call + 1: a schedule grows by one entry when the diameter doubles);
    ``return_need`` also returns the exact requirement as a 0-dim DEVICE tensor so that the caller can verify its guess later without
    stalling the launch sequence (a guess that is too small truncates the schedules that needed more)."""
    x, y = x.detach().float().contiguous(), y.detach().float().contiguous()
    P, N, _ = x.shape
    M = y.shape[1]
    both = torch.cat([x, y], 1)
    diam = (both.max(1)[0] - both.min(1)[0]).norm(dim=1).clamp_min(1e-6).double()                      # [P]
    nj = torch.ceil((math.log(blur) - diam.log()) / math.log(scaling)).clamp_min(0).long()       # len(np.arange(2 log d, 2 log blur, 2 log scaling))
    need = nj.max() + 2
    if lmax is None:
        lmax = int(need)                                                                               # the one host read
    k = torch.arange(lmax, device=x.device)[None]                                                      # [1,L]
    e_mid = (2 * diam.log()[:, None] + (k - 1).clamp_min(0) * (2 * math.log(scaling))).exp()           # d^2 scaling^(2 (k - 1))
    eps_tab = torch.where(k == 0, (diam ** 2)[:, None], e_mid)
    eps_tab = torch.where(k == (nj + 1)[:, None], torch.full_like(eps_tab, blur ** 2), eps_tab)
    eps_tab = torch.where(k > (nj + 1)[:, None], torch.zeros_like(eps_tab), eps_tab).float().t().contiguous()   # [L,P]; 0 = schedule ended
    a_log, b_log = -math.log(N), -math.log(M)
    e0 = eps_tab[0]
    # the four potentials of an iteration read only the previous iteration's: one launch for the four (csrc/optim.hip)
    g_ab, f_ba, f_aa, g_bb = _softmin_multi([(y, x, None, a_log, None), (x, y, None, b_log, None), (x, x, None, a_log, None), (y, y, None, b_log, None)],
                                            e0, False)
    for i in range(lmax):
        f_ba, g_ab, f_aa, g_bb = _softmin_multi([(x, y, g_ab, b_log, f_ba), (y, x, f_ba, a_log, g_ab), (x, x, f_aa, a_log, f_aa), (y, y, g_bb, b_log, g_bb)],
                                                eps_tab[i], True)
    last = torch.full((P,), blur ** 2, dtype=torch.float32, device=x.device)    # every schedule ends at blur^2
    f_ba_l, d_ba = _softmin_b(x, y, g_ab, b_log, last, need_grad=True)
    g_ab_l = _softmin_b(y, x, f_ba, a_log, last)
    f_aa_l, d_aa = _softmin_b(x, x, f_aa, a_log, last, need_grad=True)
    g_bb_l = _softmin_b(y, y, g_bb, b_log, last)
    loss = (f_ba_l - f_aa_l).mean(1) + (g_ab_l - g_bb_l).mean(1)
    if return_need:
        return loss, (d_ba - d_aa) / N, need
    return loss, (d_ba - d_aa) / N


class _Sinkhorn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, blur, scaling):
        loss, gx = _divergence(x.detach().float().contiguous(), y.detach().float().contiguous(), blur, scaling)
        ctx.save_for_backward(gx)
        ctx.shape = x.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (gx,) = ctx.saved_tensors
        return (g * gx).reshape(ctx.shape), None, None, None


def sinkhorn_divergence(x, y, blur=0.05, scaling=0.5):
    """x [1,N,3] or [N,3], y likewise (batch 1, as the registration loop calls it) -> scalar; differentiable w.r.t. x."""
    x2 = x.reshape(-1, 3)
    y2 = y.reshape(-1, 3)
    if not x2.is_cuda:
        raise ValueError("sinkhorn_divergence: tensors must live on the GPU (no CPU fallback)")
    return _Sinkhorn.apply(x2, y2, blur, scaling)
