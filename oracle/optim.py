"""CPU twin of the optimisation-based registration step (csrc/optim.hip) -- TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing
else); the product path never touches it.

What it restates: the loop of More_Solver._solve_pairwise_registration(optim=True)
(/root/reference/lib_more/more_solver.py:137-173, the branch /root/reference/eval_3rscan.py:381 runs per matched pair with
/root/reference/configs/more_3rscan.yaml:12-17 -- step_size.so3 0.05, n_steps 400, early_stop_threshold 10):

    g1 = LieTensor(cat(R, t), SE3); Adam([g1], lr); MultiStepLR([300, 340, 380], 0.1)                       (:137-143)
    per step: query = g1.transform(src); sdf = decoder(query; shared code);                                  (:149-152)
              loss = SmoothL1Loss()(sdf, 0) + SamplesLoss('sinkhorn', p=2)(query, tgt); backward; step       (:153-161)
              if loss < min_loss: best_g = g1 (after the step)                                               (:166-168)
              if rotmat_geodesic_distance(g1.R, init_g.R).mean() > early_stop_threshold: break               (:172-173)

**PARITY UNPINNED for the third-party parts.**  torchlie (the LieTensor parameter and what Adam does to it), geomloss and roma are
neither vendored nor installed and the reference has no test for them.  The build DEFINES:
  * manifold Adam = Adam on the 6-vector (v, omega) of the LEFT tangent space with the gradient (sum_i G_i, sum_i q_i x G_i),
    G = d loss / d (g . src), retraction g <- exp(-step) g;
  * SmoothL1 with torch's defaults (beta = 1, mean);
  * the Sinkhorn divergence of oracle/sinkhorn.py;
  * geodesic angle in RADIANS compared with the configured number (roma returns radians and the reference compares them with "10").
This file follows those definitions with plain torch on the CPU (dense ops, torch autograd for the decoder and the Sinkhorn term) so
that every kernel of csrc/optim.hip has an independent implementation of the SAME definition to be compared with; the reference-owned
pieces it leans on (decoder, SE(3) helpers) are the pinned ones of oracle/net.py / oracle/more.py.
"""
import math

import torch


def se3_transform(g, src):
    """g [P,3,4] = (R | t), src [P,N,3] -> R src + t  (more_solver.py:149 g1.transform(src_pc))."""
    return src @ g[:, :, :3].transpose(1, 2) + g[:, None, :, 3]


def smooth_l1(sdf):
    """torch.nn.SmoothL1Loss() of (sdf, 0) per pair (more_solver.py:144,153): -> (loss [P], d loss / d sdf [P,N])."""
    x = sdf.detach().clone().requires_grad_(True)
    per = torch.nn.functional.smooth_l1_loss(x, torch.zeros_like(x), reduction="none").mean(1)
    per.sum().backward()
    return per.detach(), x.grad


def se3_exp(xi):
    """exp of the twist (v, omega) [6] -> [3,4]: Rodrigues + the left Jacobian on the translation; first order below 1e-6 rad."""
    v, w = xi[:3], xi[3:]
    th = float(w.norm())
    K = torch.zeros(3, 3, dtype=xi.dtype)
    K[0, 1], K[0, 2], K[1, 0], K[1, 2], K[2, 0], K[2, 1] = -w[2], w[1], w[2], -w[0], -w[1], w[0]
    eye = torch.eye(3, dtype=xi.dtype)
    if th < 1e-6:
        R, V = eye + K, eye + 0.5 * K
    else:
        a, b, c = math.sin(th) / th, (1 - math.cos(th)) / th ** 2, (th - math.sin(th)) / th ** 3
        R, V = eye + a * K + b * (K @ K), eye + b * K + c * (K @ K)
    return torch.cat([R, (V @ v)[:, None]], 1)


class Se3AdamState:
    """The per-pair state ls_se3_adam_step_f32 carries (ops.Se3Adam)."""

    def __init__(self, g0, src, stop_angle, betas=(0.9, 0.999), eps=1e-8):
        self.src = src.clone()
        self.g = g0.clone()
        P = g0.shape[0]
        self.m1 = torch.zeros(P, 6, dtype=g0.dtype)
        self.m2 = torch.zeros(P, 6, dtype=g0.dtype)
        self.min_loss = torch.full((P,), 100.0, dtype=g0.dtype)          # more_solver.py:141
        self.best_g = g0.clone()
        self.init_R = g0[:, :, :3].clone()
        self.active = torch.ones(P, dtype=torch.bool)
        self.query = se3_transform(self.g, self.src)
        self.betas, self.eps, self.stop_angle, self.step_no = betas, eps, float(stop_angle), 0


def tangent_gradient(query, G):
    """d loss / d (v, omega) of g <- exp((v, omega)) g at 0 from the point gradients G = d loss / d query: (sum G, sum q x G)."""
    return torch.cat([G.sum(1), torch.cross(query, G, dim=-1).sum(1)], 1)


def se3_adam_step(st, grad_query, loss, lr):
    """One step for every ACTIVE pair (a stopped pair is frozen: the reference leaves its loop, :172-173)."""
    b1, b2 = st.betas
    t = st.step_no + 1
    gr = tangent_gradient(st.query, grad_query)
    for p in range(st.g.shape[0]):
        if not st.active[p]:
            continue
        st.m1[p] = b1 * st.m1[p] + (1 - b1) * gr[p]
        st.m2[p] = b2 * st.m2[p] + (1 - b2) * gr[p] * gr[p]
        step = -(lr * (st.m1[p] / (1 - b1 ** t)) / ((st.m2[p] / (1 - b2 ** t)).sqrt() + st.eps))
        E = se3_exp(step)
        gn = torch.cat([E[:, :3] @ st.g[p, :, :3], E[:, :3] @ st.g[p, :, 3:] + E[:, 3:]], 1)
        st.g[p] = gn
        if loss[p] < st.min_loss[p]:                                   # the snapshot is the pose AFTER the step (:166-168)
            st.min_loss[p] = loss[p]
            st.best_g[p] = gn
        cosang = ((gn[:, :3] * st.init_R[p]).sum() - 1) / 2            # trace(R R0^T)
        if math.acos(min(1.0, max(-1.0, float(cosang)))) > st.stop_angle:
            st.active[p] = False
        st.query[p] = se3_transform(gn[None], st.src[p][None])[0]
    st.step_no += 1


def lr_at(step, lr0, milestones=(300, 340, 380), gamma=0.1):
    """MultiStepLR (more_solver.py:143): the rate the optimizer uses AT step `step` (0-based)."""
    return lr0 * gamma ** sum(step >= m for m in milestones)


def registration_loop(dec_w, dec_cfg, code, src, tgt, g0, n_steps, lr0, stop_angle, trace=None):
    """The refinement loop for P pairs (each exactly as if alone): code = the shared code dict (P rows), src / tgt [P,N,3] / [P,M,3],
    g0 [P,3,4].  Returns the final state; trace (list) receives (g, loss) per step."""
    from . import net, sinkhorn
    st = Se3AdamState(g0, src, stop_angle)
    P = g0.shape[0]
    for i in range(n_steps):
        q = st.query.detach().clone().requires_grad_(True)
        sdf = net.field_query_with_grad(dec_w, dec_cfg, q, code)
        sdf_loss = torch.nn.functional.smooth_l1_loss(sdf, torch.zeros_like(sdf), reduction="none").mean(1)
        ot = torch.stack([sinkhorn.sinkhorn_divergence(q[p], tgt[p]) for p in range(P)])
        loss = sdf_loss + ot
        loss.sum().backward()
        se3_adam_step(st, q.grad, loss.detach(), lr_at(i, lr0))
        if trace is not None:
            trace.append((st.g.clone(), loss.detach().clone()))
        if not bool(st.active.any()):
            break
    return st
