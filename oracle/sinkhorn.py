"""The same debiased Sinkhorn divergence as livingscenes_amd/sinkhorn.py on torch-CPU with autograd -- TEST INFRASTRUCTURE ONLY.

**Parity UNPINNED**: the reference calls geomloss.SamplesLoss(loss='sinkhorn', p=2) (/root/reference/lib_more/more_solver.py:146),
geomloss is neither vendored nor installed, and the reference has no test for it.  Both files restate geomloss' published
algorithm from memory; this one exists to check the HIP softmin kernel and the loop around it against an independent
implementation of the SAME definition (dense cost matrices, torch.logsumexp, gradients from autograd).
"""
import math

import numpy as np
import torch


def _softmin(eps, C, h):
    return -eps * (h[None, :] - C / eps).logsumexp(1)


def sinkhorn_divergence(x, y, blur=0.05, scaling=0.5, p=2):
    x, y = x.reshape(-1, 3), y.reshape(-1, 3)
    N, M = x.shape[0], y.shape[0]
    xd, yd = x.detach(), y.detach()

    def cost(a, b):
        return 0.5 * ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    both = torch.cat([xd, yd], 0)
    diameter = max(float((both.max(0)[0] - both.min(0)[0]).norm()), 1e-6)
    eps_list = ([diameter ** p] + [float(np.exp(e)) for e in np.arange(p * math.log(diameter), p * math.log(blur), p * math.log(scaling))]
                + [blur ** p])
    a_log = torch.full((N,), -math.log(N))
    b_log = torch.full((M,), -math.log(M))
    with torch.no_grad():
        Cxy, Cyx, Cxx, Cyy = cost(xd, yd), cost(yd, xd), cost(xd, xd), cost(yd, yd)
        eps = eps_list[0]
        g_ab, f_ba = _softmin(eps, Cyx, a_log), _softmin(eps, Cxy, b_log)
        f_aa, g_bb = _softmin(eps, Cxx, a_log), _softmin(eps, Cyy, b_log)
        for eps in eps_list:
            ft_ba = _softmin(eps, Cxy, b_log + g_ab / eps)
            gt_ab = _softmin(eps, Cyx, a_log + f_ba / eps)
            ft_aa = _softmin(eps, Cxx, a_log + f_aa / eps)
            gt_bb = _softmin(eps, Cyy, b_log + g_bb / eps)
            f_ba, g_ab = 0.5 * (f_ba + ft_ba), 0.5 * (g_ab + gt_ab)
            f_aa, g_bb = 0.5 * (f_aa + ft_aa), 0.5 * (g_bb + gt_bb)
    f_ba_l = _softmin(eps, cost(x, yd), b_log + g_ab / eps)
    g_ab_l = _softmin(eps, cost(yd, xd), a_log + f_ba / eps)
    f_aa_l = _softmin(eps, cost(x, xd), a_log + f_aa / eps)
    g_bb_l = _softmin(eps, cost(yd, yd), b_log + g_bb / eps)
    return (f_ba_l - f_aa_l).mean() + (g_ab_l - g_bb_l).mean()
