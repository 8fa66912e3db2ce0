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
