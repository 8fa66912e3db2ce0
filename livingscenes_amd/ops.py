"""Tensor-level wrappers over the C ABI (include/livingscenes_hip.h).  Inputs/outputs are HIP torch tensors;
torch only provides device memory and the current stream.  No CPU fallback anywhere (see _lib.ptr)."""
import ctypes
import os

import torch

from . import _lib
from ._lib import call, check, load, ptr, stream_ptr

DEFAULT_FLAGS = 0  # canonical arithmetic: separately rounded multiply/add (oracle contract=0)


def _f32(t):
    return t.contiguous().float() if (t.dtype != torch.float32 or not t.is_contiguous()) else t


def _scratch(nbytes, device):
    """Caller-owned scratch for one operator call (torch's caching allocator: stream-ordered reuse, no hipMalloc per call)."""
    return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None


def knn(dst, src, K=16, dst_rows=None, flags=DEFAULT_FLAGS, return_dist=False, seeds=None):
    """dst [B,Nd',3,C], src [B,Ns,3,C] -> idx [B,Nd,K] int32 (Nd = dst_rows.shape[1] if given).
    seeds [B,Nd,16] int32: optional candidate hints (result independent of them)."""
    dst, src = _f32(dst), _f32(src)
    B, dst_n, _, C = dst.shape
    Ns = src.shape[1]
    Nd = dst_rows.shape[1] if dst_rows is not None else dst_n
    idx = torch.empty(B, Nd, K, dtype=torch.int32, device=src.device)
    dist = torch.empty(B, Nd, K, dtype=torch.float32, device=src.device) if return_dist else None
    if seeds is not None:
        seeds = seeds.to(torch.int32).contiguous()
        assert seeds.shape == (B, Nd, 16)
    ws = _scratch(load().ls_knn_workspace_bytes(B, Nd, dst_n, Ns, C, int(seeds is not None), flags), src.device)
    call(src.device, "ls_knn_f32", ptr(dst), ptr(src), ptr(dst_rows), ptr(seeds), B, Nd, dst_n, Ns, C, K, flags, ptr(idx), ptr(dist),
                            ptr(ws), 0 if ws is None else ws.numel(), stream_ptr(src.device))
    return (idx, dist) if return_dist else idx


def fps(pts, K, lengths=None, flags=DEFAULT_FLAGS, return_points=False):
    """pts [B,N,3] -> idx [B,K] int32 (and the gathered points [B,K,3])."""
    pts = _f32(pts)
    B, N, _ = pts.shape
    idx = torch.empty(B, K, dtype=torch.int32, device=pts.device)
    out = torch.empty(B, K, 3, dtype=torch.float32, device=pts.device) if return_points else None
    if lengths is not None:
        lengths = lengths.to(device=pts.device, dtype=torch.int32).contiguous()
    ws = _scratch(load().ls_fps_workspace_bytes(B, N, K), pts.device)
    call(pts.device, "ls_fps_f32", ptr(pts), ptr(lengths), B, N, K, flags, ptr(idx), ptr(out), ptr(ws), 0 if ws is None else ws.numel(),
         stream_ptr(pts.device))
    return (idx, out) if return_points else idx


def gemm(A, W, bias=None, relu=False):
    """act(A[M,K] @ W[N,K]^T + bias) -> [M,N]."""
    A, W = _f32(A), _f32(W)
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    ws = _scratch(load().ls_gemm_workspace_bytes(M, N, K), A.device)
    call(A.device, "ls_gemm_f32", ptr(A), K, ptr(W), K, ptr(bias), ptr(out), N, M, N, K, int(relu), ptr(ws), 0 if ws is None else ws.numel(),
                             stream_ptr(A.device))
    return out


def rowmax(X):
    """max_k |X[r, k]| -> [rows]: the operand range ls_gemm_f32_ex takes for a weight matrix (once) or an input."""
    X = _f32(X)
    out = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
    call(X.device, "ls_rowmax_f32", ptr(X), X.shape[0], X.shape[1], X.shape[1], ptr(out), stream_ptr(X.device))
    return out


def presplit_w(W, w_rowmax):
    """ls_gemm_presplit_w_f32: both f16 pieces of the range-scaled rows of a weight matrix, for ``gemm_chain(..., w_planes=)``.  None when
    no kernel reads planes at this K (K < 512 or K % 32 != 0)."""
    W = _f32(W)
    N, K = W.shape
    nb = load().ls_gemm_w_planes_bytes(N, K)
    if nb == 0:
        return None
    planes = torch.empty(nb, dtype=torch.uint8, device=W.device)
    call(W.device, "ls_gemm_presplit_w_f32", ptr(W), K, N, K, ptr(w_rowmax), ptr(planes), nb, stream_ptr(W.device))
    return planes


def gemm_chain(A, W, bias=None, relu=False, a_rowmax=None, w_rowmax=None, want_rowmax=True, w_planes=None):
    """ls_gemm_f32_ex: the GEMM of ``gemm`` for a chain of layers -- takes the row maxima of its operands (``a_rowmax`` [M, parts] from
    the previous call, ``w_rowmax`` [N] from ``rowmax(W)``) and returns (out, out_rowmax [M, parts']) for the next one.  Never splits
    K (a row's result does not depend on the other rows of the call).  ``w_planes`` (``presplit_w(W, w_rowmax)``): ls_gemm_f32_planes,
    the same result without re-splitting W in every workgroup."""
    A, W = _f32(A), _f32(W)
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    parts = load().ls_gemm_rowmax_parts(N)
    orm = torch.empty(M, parts, dtype=torch.float32, device=A.device) if want_rowmax else None
    a_parts = 0 if a_rowmax is None else (1 if a_rowmax.dim() == 1 else a_rowmax.shape[1])
    if w_planes is not None:
        call(A.device, "ls_gemm_f32_planes", ptr(A), K, ptr(W), K, ptr(w_planes), ptr(bias), ptr(out), N, M, N, K, int(relu), ptr(a_rowmax),
             a_parts, ptr(w_rowmax), ptr(orm), stream_ptr(A.device))
        return out, orm
    call(A.device, "ls_gemm_f32_ex", ptr(A), K, ptr(W), K, ptr(bias), ptr(out), N, M, N, K, int(relu), ptr(a_rowmax), a_parts, ptr(w_rowmax),
         ptr(orm), None, 0, stream_ptr(A.device))
    return out, orm


def encode_prologue(x):
    """x [B,3,N] -> (pts [B,N,3], centroid [B,3], scale0 [B])."""
    x = _f32(x)
    B, _, N = x.shape
    pts = torch.empty(B, N, 3, dtype=torch.float32, device=x.device)
    cen = torch.empty(B, 3, dtype=torch.float32, device=x.device)
    sc = torch.empty(B, dtype=torch.float32, device=x.device)
    call(x.device, "ls_encode_prologue_f32", ptr(x), B, N, ptr(pts), ptr(cen), ptr(sc), stream_ptr(x.device))
    return pts, cen, sc


def cosine_scores(m0, m1):
    m0, m1 = _f32(m0), _f32(m1)
    n, D = m0.shape
    m = m1.shape[0]
    S = torch.empty(n, m, dtype=torch.float32, device=m0.device)
    ws = _scratch(load().ls_cosine_scores_workspace_bytes(n, m), m0.device)
    call(m0.device, "ls_cosine_scores_f32", ptr(m0), ptr(m1), n, m, D, ptr(S), ptr(ws), ws.numel(), stream_ptr(m0.device))
    return S


def greedy_match(scores):
    """scores [n,m] -> (matches0 [n], matches1 [m]) int64.  (The large-problem kernel normalises its input in place: copied then;
    the single-wave kernel of n * m <= 1024 keeps the matrix in registers and leaves `scores` untouched.)"""
    S = _f32(scores)
    n, m = S.shape
    if n * m > 1024:
        S = S.clone()
    m0 = torch.empty(n, dtype=torch.int64, device=S.device)
    m1 = torch.empty(m, dtype=torch.int64, device=S.device)
    call(S.device, "ls_greedy_match_f32", ptr(S), n, m, ptr(m0), ptr(m1), stream_ptr(S.device))
    return m0, m1


def nn_match(scores):
    """scores [n,m] -> mutual-nearest-neighbour (matches0 [n], matches1 [m]) int64 (matcher_new.py:89-98), one launch."""
    S = _f32(scores)
    n, m = S.shape
    m0 = torch.empty(n, dtype=torch.int64, device=S.device)
    m1 = torch.empty(m, dtype=torch.int64, device=S.device)
    call(S.device, "ls_nn_match_f32", ptr(S), n, m, ptr(m0), ptr(m1), stream_ptr(S.device))
    return m0, m1


def sinkhorn_match(scores, score_divisor, alpha=1.0, iters=100, match_threshold=0.0):
    """scores [n,m] -> (matches0 [n], matches1 [m]) int64: log-space optimal transport with a dustbin + mutual arg-maxes (matcher_new.py:20-71),
    the whole loop in one launch."""
    S = _f32(scores)
    n, m = S.shape
    m0 = torch.empty(n, dtype=torch.int64, device=S.device)
    m1 = torch.empty(m, dtype=torch.int64, device=S.device)
    call(S.device, "ls_sinkhorn_match_f32", ptr(S), n, m, float(score_divisor), float(alpha), int(iters), float(match_threshold), ptr(m0), ptr(m1),
         stream_ptr(S.device))
    return m0, m1


def kabsch(x1, x2, weights=None, return_flags=False, raw_weights=False):
    """x1,x2 [b,n,3] -> R [b,3,3], t [b,3,1], res [b,n] (, status [b] int32: _lib.KABSCH_*).  raw_weights: use `weights` as
    they are instead of normalising them (pose_estimation.py:52-54)."""
    x1, x2 = _f32(x1), _f32(x2)
    b, n, _ = x1.shape
    if weights is not None:
        weights = _f32(weights)
    R = torch.empty(b, 3, 3, dtype=torch.float32, device=x1.device)
    t = torch.empty(b, 3, dtype=torch.float32, device=x1.device)
    res = torch.empty(b, n, dtype=torch.float32, device=x1.device)
    fl = torch.empty(b, dtype=torch.int32, device=x1.device)
    call(x1.device, "ls_kabsch_batched_f32", ptr(x1), ptr(x2), ptr(weights), b, n, _lib.FLAG_KABSCH_RAW_WEIGHTS if raw_weights else 0,
                                       ptr(R), ptr(t), ptr(res), ptr(fl),
                                       stream_ptr(x1.device))
    return (R, t.unsqueeze(2), res, fl) if return_flags else (R, t.unsqueeze(2), res)


def kabsch_codes(z1, t1, z2, t2, sel2=None, sel1=None, want_res=False):
    """Kabsch on the pseudo-points z + t of two code sets (more_solver.py:114-116) in ONE launch: z [n,c,3], t [n,1,3] or [n,3];
    sel2 / sel1 (int64 [b], optional): problem p pairs set sel1[p] (default p) of side 1 with set sel2[p] of side 2 -- e.g. matches0
    (negative = unmatched -> set 0, as matches0.clamp(min=0)).  -> R [b,3,3], t [b,3,1] (, res [b,c])."""
    z1, z2 = _f32(z1), _f32(z2)
    t1, t2 = _f32(t1.reshape(-1, 3)), _f32(t2.reshape(-1, 3))
    b = (sel1 if sel1 is not None else sel2).shape[0] if (sel1 is not None or sel2 is not None) else z1.shape[0]
    n = z1.shape[1]
    R = torch.empty(b, 3, 3, dtype=torch.float32, device=z1.device)
    t = torch.empty(b, 3, dtype=torch.float32, device=z1.device)
    res = torch.empty(b, n, dtype=torch.float32, device=z1.device) if want_res else None
    if sel1 is not None:
        sel1 = sel1.contiguous().long()
    if sel2 is not None:
        sel2 = sel2.contiguous().long()
    call(z1.device, "ls_kabsch_codes_f32", ptr(z1), ptr(t1), ptr(sel1), ptr(z2), ptr(t2), ptr(sel2), b, n, ptr(R), ptr(t), ptr(res), None,
         stream_ptr(z1.device))
    return (R, t.unsqueeze(2), res) if want_res else (R, t.unsqueeze(2))


def kabsch_residual_matrix(src, tgt):
    """src [n,P,3], tgt [m,P,3] -> mean residual [n,m]."""
    src, tgt = _f32(src), _f32(tgt)
    n, P, _ = src.shape
    m = tgt.shape[0]
    res = torch.empty(n, m, dtype=torch.float32, device=src.device)
    call(src.device, "ls_kabsch_residual_matrix_f32", ptr(src), ptr(tgt), n, m, P, ptr(res), stream_ptr(src.device))
    return res


def icp(X, Y, R0, T0, max_iter=100, rel_rmse_thr=1e-6, flags=DEFAULT_FLAGS):
    """Row-vector convention Xt = X R + T.  X [b,n,3], Y [b,m,3], R0 [b,3,3], T0 [b,3] -> R, T, rmse [b], iters [b]."""
    X, Y, R0, T0 = _f32(X), _f32(Y), _f32(R0), _f32(T0)
    b, n, _ = X.shape
    m = Y.shape[1]
    R = torch.empty(b, 3, 3, dtype=torch.float32, device=X.device)
    T = torch.empty(b, 3, dtype=torch.float32, device=X.device)
    rmse = torch.empty(b, dtype=torch.float32, device=X.device)
    iters = torch.empty(b, dtype=torch.int32, device=X.device)
    ws = torch.empty(load().ls_icp_workspace_bytes(b, n), dtype=torch.uint8, device=X.device)
    call(X.device, "ls_icp_f32", ptr(X), ptr(Y), ptr(R0), ptr(T0), b, n, m, max_iter, rel_rmse_thr, flags, ptr(R), ptr(T),
                            ptr(rmse), ptr(iters), ptr(ws), ws.numel(), stream_ptr(X.device))
    return R, T, rmse, iters


def se3_transform(g, src):
    """g [P,3,4] (R | t), src [P,N,3] -> g . src [P,N,3]."""
    g, src = _f32(g), _f32(src)
    P, N, _ = src.shape
    out = torch.empty_like(src)
    call(src.device, "ls_se3_transform_f32", ptr(g), ptr(src), P, N, ptr(out), stream_ptr(src.device))
    return out


def smooth_l1(sdf, loss=None):
    """Per-row mean SmoothL1(sdf, 0) of sdf [P,N] -> (loss [P] (added to `loss` if given), d loss / d sdf [P,N])."""
    sdf = _f32(sdf)
    P, N = sdf.shape
    acc = loss is not None
    if loss is None:
        loss = torch.empty(P, dtype=torch.float32, device=sdf.device)
    grad = torch.empty_like(sdf)
    call(sdf.device, "ls_smooth_l1_f32", ptr(sdf), P, N, int(acc), ptr(loss), ptr(grad), stream_ptr(sdf.device))
    return loss, grad


def mse(sdf, min_loss=None, improved=None):
    """Per-row MSELoss(sdf, 0) of sdf [P,N] -> (loss [P], d loss / d sdf [P,N]); min_loss [P] float / improved [P] int32 (optional) are
    updated in place where the loss improved (More_Solver._optimize_code's bookkeeping, more_solver.py:219-221)."""
    sdf = _f32(sdf)
    P, N = sdf.shape
    loss = torch.empty(P, dtype=torch.float32, device=sdf.device)
    grad = torch.empty_like(sdf)
    call(sdf.device, "ls_mse_f32", ptr(sdf), P, N, ptr(loss), ptr(grad), ptr(min_loss), ptr(improved), stream_ptr(sdf.device))
    return loss, grad


class Adam:
    """torch.optim.Adam(params with per-group lr) on the device: ONE launch per step for up to four tensors (csrc/optim.hip:
    adam_multi_kernel, torch's operation order).  params: list of (tensor updated in place, lr)."""

    def __init__(self, params, betas=(0.9, 0.999), eps=1e-8):
        assert 1 <= len(params) <= 4
        self.params = [(p, float(lr)) for p, lr in params]
        for p, _ in self.params:
            assert p.is_contiguous() and p.dtype == torch.float32
        self.m = [torch.zeros_like(p) for p, _ in self.params]
        self.v = [torch.zeros_like(p) for p, _ in self.params]
        self.betas, self.eps, self.step_no = betas, eps, 0

    def step(self, grads, lr_scale=1.0):
        groups = (_lib.AdamGroup * len(self.params))()
        keep = []
        for i, ((p, lr), g) in enumerate(zip(self.params, grads)):
            g = _f32(g.reshape(p.shape))
            keep.append(g)
            groups[i] = _lib.AdamGroup(p.data_ptr(), g.data_ptr(), self.m[i].data_ptr(), self.v[i].data_ptr(), p.numel(), lr * lr_scale)
        call(self.params[0][0].device, "ls_adam_step_f32", groups, len(self.params), self.betas[0], self.betas[1], self.eps, self.step_no,
             stream_ptr(self.params[0][0].device))
        self.step_no += 1


class Se3Adam:
    """State of the batched SE(3) Adam of the optimisation-based registration (csrc/optim.hip: se3_adam_step_kernel)."""

    def __init__(self, g0, src, stop_angle, betas=(0.9, 0.999), eps=1e-8):
        self.src = _f32(src)
        self.g = _f32(g0).clone()
        P = self.g.shape[0]
        dev = self.g.device
        self.m1 = torch.zeros(P, 6, device=dev)
        self.m2 = torch.zeros(P, 6, device=dev)
        self.min_loss = torch.full((P,), 100.0, device=dev)       # more_solver.py:141
        self.best_g = self.g.clone()
        self.init_R = self.g[:, :, :3].contiguous().clone()
        self.active = torch.ones(P, dtype=torch.int32, device=dev)
        self.query = se3_transform(self.g, self.src)
        self.betas, self.eps, self.stop_angle, self.step_no = betas, eps, float(stop_angle), 0

    def step(self, grad_query, loss, lr):
        P, N, _ = self.src.shape
        call(self.src.device, "ls_se3_adam_step_f32", ptr(self.src), ptr(_f32(grad_query)), ptr(_f32(loss)), P, N, float(lr), self.betas[0],
             self.betas[1], self.eps, self.step_no, self.stop_angle, ptr(self.g), ptr(self.m1), ptr(self.m2), ptr(self.min_loss),
             ptr(self.best_g), ptr(self.init_R), ptr(self.active), ptr(self.query), stream_ptr(self.src.device))
        self.step_no += 1


class HipModel:
    """Owner of an ls_model_t (device-resident packed weights) with encode / sdf_decode entry points."""

    def __init__(self, desc, blob, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LsError("HipModel needs a HIP device: the MI355X path has no CPU fallback")
        self.desc = desc
        self._h = ctypes.c_void_p()
        self._ws = {}
        with torch.cuda.device(self.device):
            check(load().ls_model_create(ctypes.byref(desc), blob.ctypes.data_as(ctypes.c_void_p), ctypes.byref(self._h)),
                  "ls_model_create")

    def close(self):
        if self._h:
            load().ls_model_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _workspace(self, nbytes):
        """Grow-only scratch, one buffer per STREAM the handle is used on (two streams driving one handle must not share it)."""
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return ws

    def get_option(self, option):
        """ls_model_get_option (_lib.OPT_*): the handle's current value, read from the library (never mirrored in Python)."""
        v = ctypes.c_int(0)
        check(load().ls_model_get_option(self._h, int(option), ctypes.byref(v)), "ls_model_get_option")
        return int(v.value)

    def set_option(self, option, value):
        """ls_model_set_option (_lib.OPT_*) -> the value the option had before (asked of the library), so that a temporary change can be
        undone without clobbering a caller's setting or the environment's default."""
        prev = self.get_option(option)
        check(load().ls_model_set_option(self._h, int(option), int(value)), "ls_model_set_option")
        return prev

    def profile_begin(self):
        check(load().ls_profile_begin(self._h), "ls_profile_begin")

    def profile_end(self):
        """-> list of dicts {kind, layer, launches, total_ms} (hipEvent-bracketed per launch)."""
        buf = (_lib.ProfileEntry * 256)()
        n = ctypes.c_int(0)
        check(load().ls_profile_end(self._h, buf, 256, ctypes.byref(n)), "ls_profile_end")
        return [dict(kind=_lib.KERNEL_KINDS[buf[i].kind], layer=buf[i].layer, launches=buf[i].launches, total_ms=buf[i].total_ms)
                for i in range(n.value)]

    def profile_knn_stats(self):
        """ls_profile_knn_stats -> {layer: (candidates given a canonical distance beyond the hints, queries)} of the profiled encode calls."""
        buf = (ctypes.c_ulonglong * 16)()
        check(load().ls_profile_knn_stats(self._h, buf, 8), "ls_profile_knn_stats")
        return {i: (int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(8) if buf[2 * i + 1]}

    def n_levels(self):
        return [int(self.desc.down_factor[i]) for i in range(self.desc.num_layers)]

    def encode(self, x, pre_normalised=False, flags=DEFAULT_FLAGS, trace=False):
        """x [B,3,N] -> (z_so3 [B,C,3], z_inv [B,C], s [B], t [B,3]) (+ per-layer knn / fps index lists)."""
        x = _f32(x)
        B, three, N = x.shape
        assert three == 3
        d = self.desc
        C = d.c_dim
        dev = x.device
        z_so3 = torch.empty(B, C, 3, dtype=torch.float32, device=dev)
        z_inv = torch.empty(B, C, dtype=torch.float32, device=dev)
        s = torch.empty(B, dtype=torch.float32, device=dev)
        t = torch.empty(B, 3, dtype=torch.float32, device=dev)
        need = load().ls_encoder_workspace_bytes(self._h, B, N)
        if need == 0:
            raise _lib.LsError("ls_encoder_workspace_bytes: " + load().ls_last_error().decode())
        ws = self._workspace(need)
        tk = tf = None
        nds, fps_n = [], []
        cur = N
        for i in range(d.num_layers):
            if d.down_factor[i] > 1:
                cur //= d.down_factor[i]
                fps_n.append(cur)
            nds.append(cur)
        if trace:
            tk = torch.empty(B * sum(nds) * 16, dtype=torch.int32, device=dev)
            tf = torch.empty(max(1, B * sum(fps_n)), dtype=torch.int32, device=dev)
        call(dev, "ls_encode", self._h, ptr(x), B, N, int(pre_normalised), flags, ptr(z_so3), ptr(z_inv), ptr(s), ptr(t),
                               ptr(tk), ptr(tf), ptr(ws), ws.numel(), stream_ptr(dev))
        if not trace:
            return z_so3, z_inv, s, t
        knn_l, fps_l, o = [], [], 0
        for nd in nds:
            knn_l.append(tk[o:o + B * nd * 16].view(B, nd, 16))
            o += B * nd * 16
        o = 0
        for nf in fps_n:
            fps_l.append(tf[o:o + B * nf].view(B, nf))
            o += B * nf
        return z_so3, z_inv, s, t, knn_l, fps_l

    # ---- the encoder's layer operators on their own (same code path as ls_encode)
    def edgeconv(self, layer, src_f, knn, dst_rows=None, _ws=None):
        """Edge-conv layer `layer` without its residual global conv: src_f [B,Ns,3,C_in] (layer 0: cloud [B,Ns,3]), knn [B,Nd,16]
        int32, dst_rows [B,Nd] int32 or None -> [B,Nd,3,C_out]."""
        src_f = _f32(src_f)
        knn = knn.to(torch.int32).contiguous()
        B, Ns = src_f.shape[0], src_f.shape[1]
        Nd = knn.shape[1]
        if dst_rows is not None:
            dst_rows = dst_rows.to(torch.int32).contiguous()
        Co = int(self.desc.feat_dim[layer])
        out = torch.empty(B, Nd, 3, Co, dtype=torch.float32, device=src_f.device)
        ws = _ws if _ws is not None else _scratch(load().ls_vn_edgeconv_workspace_bytes(self._h, layer, B, Ns, Nd, int(dst_rows is not None)), src_f.device)
        fn = "ls_vn_edgeconv_attn_f32" if layer >= self.desc.atten_start_layer else "ls_vn_edgeconv_pool_f32"
        call(src_f.device, fn, self._h, layer, ptr(src_f), ptr(knn), ptr(dst_rows), B, Ns, Nd, ptr(out), ptr(ws), 0 if ws is None else ws.numel(),
             stream_ptr(src_f.device))
        return out

    def vn_lna_global(self, layer, f):
        """Residual global conv of `layer`: f [B,N,3,C] -> VecLNA_G(cat(f, mean_n f)) [B,N,3,C]."""
        f = _f32(f)
        B, N = f.shape[0], f.shape[1]
        out = torch.empty_like(f)
        ws = _scratch(load().ls_vn_lna_workspace_bytes(self._h, layer, B, N), f.device)
        call(f.device, "ls_vn_lna_f32", self._h, layer, ptr(f), B, N, ptr(out), ptr(ws), ws.numel(), stream_ptr(f.device))
        return out

    def encoder_tail(self, f, centroid=None, scale0=None):
        """f [B,NP,3,C_last] -> (z_so3 [B,c,3], z_inv [B,c], s [B], t [B,3])."""
        f = _f32(f)
        B, NP = f.shape[0], f.shape[1]
        C, dev = self.desc.c_dim, f.device
        z_so3 = torch.empty(B, C, 3, dtype=torch.float32, device=dev)
        z_inv = torch.empty(B, C, dtype=torch.float32, device=dev)
        s, t = torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, 3, dtype=torch.float32, device=dev)
        ws = _scratch(load().ls_encoder_tail_workspace_bytes(self._h, B, NP), dev)
        call(dev, "ls_encoder_tail_f32", self._h, ptr(f), ptr(None if centroid is None else _f32(centroid)),
                                         ptr(None if scale0 is None else _f32(scale0)), B, NP, ptr(z_so3), ptr(z_inv), ptr(s), ptr(t),
                                         ptr(ws), ws.numel(), stream_ptr(dev))
        return z_so3, z_inv, s, t

    def sdf_decode(self, query, z_so3, z_inv, s, t, max_ws_bytes=2 << 30):
        """query [B,M,3] + code -> sdf [B,M]; queries are processed in chunks that keep the workspace bounded."""
        query = _f32(query)
        B, M, _ = query.shape
        z_so3, z_inv, s, t = _f32(z_so3), _f32(z_inv), _f32(s), _f32(t.reshape(B, 3))
        sdf = torch.empty(B, M, dtype=torch.float32, device=query.device)
        per_q = 2 * self.desc.dec_width * 4 * B
        chunk = max(64, min(M, int(max_ws_bytes // max(per_q, 1))))
        for m0 in range(0, M, chunk):
            mc = min(chunk, M - m0)
            q = query[:, m0:m0 + mc].contiguous()
            out = sdf if mc == M else torch.empty(B, mc, dtype=torch.float32, device=query.device)
            need = load().ls_sdf_workspace_bytes(self._h, B, mc)
            ws = self._workspace(need)
            call(query.device, "ls_sdf_decode", self._h, ptr(q), ptr(z_so3), ptr(z_inv), ptr(s), ptr(t), B, mc, ptr(out), ptr(ws),
                                       ws.numel(), stream_ptr(query.device))
            if mc != M:
                sdf[:, m0:m0 + mc] = out
        return sdf

    def sdf_decode_rows(self, query, row_inst, z_so3, z_inv, s, t, max_rows=1 << 20):
        """Ragged batch: query [R,3], row_inst [R] int32 (instance of each row; rows of an instance contiguous) -> sdf [R]."""
        query = _f32(query)
        R = query.shape[0]
        B = z_inv.shape[0]
        row_inst = row_inst.to(torch.int32).contiguous()
        z_so3, z_inv, s, t = _f32(z_so3), _f32(z_inv), _f32(s), _f32(t.reshape(B, 3))
        sdf = torch.empty(R, dtype=torch.float32, device=query.device)
        for r0 in range(0, R, max_rows):
            rc = min(max_rows, R - r0)
            q, ri, out = query[r0:r0 + rc], row_inst[r0:r0 + rc], sdf[r0:r0 + rc]
            need = load().ls_sdf_rows_workspace_bytes(self._h, B, rc)
            ws = self._workspace(need)
            call(query.device, "ls_sdf_decode_rows", self._h, ptr(q), ptr(ri), ptr(z_so3), ptr(z_inv), ptr(s), ptr(t), B, rc, ptr(out), ptr(ws),
                                            ws.numel(), stream_ptr(query.device))
        return sdf

    def sdf_decode_train(self, query, z_so3, z_inv, s, t):
        """Forward that keeps the activations: -> (sdf [B,M], saved) where ``saved`` feeds sdf_backward (its workspace is
        private to the call, so several graphs can be alive at once)."""
        query = _f32(query)
        B, M, _ = query.shape
        z_so3, z_inv, s, t = _f32(z_so3), _f32(z_inv), _f32(s), _f32(t.reshape(B, 3))
        sdf = torch.empty(B, M, dtype=torch.float32, device=query.device)
        need = load().ls_sdf_train_workspace_bytes(self._h, B, M)
        ws = torch.empty(need, dtype=torch.uint8, device=query.device)
        call(query.device, "ls_sdf_decode_train", self._h, ptr(query), ptr(z_so3), ptr(z_inv), ptr(s), ptr(t), B, M, ptr(sdf), ptr(ws),
                                         ws.numel(), stream_ptr(query.device))
        return sdf, (query, z_so3, z_inv, s, t, sdf, ws)

    def sdf_backward(self, saved, grad_sdf, need_query_grad=True, need_code_grad=True):
        """-> (grad_query [B,M,3] or None, grad_z_so3 [B,c,3] or None, grad_z_inv [B,c] or None, grad_s [B], grad_t [B,3])."""
        query, z_so3, z_inv, s, t, sdf, ws = saved
        B, M, _ = query.shape
        dev = query.device
        g = _f32(grad_sdf).reshape(B, M)
        gq = torch.empty(B, M, 3, dtype=torch.float32, device=dev) if need_query_grad else None
        gso3, ginv = (torch.empty_like(z_so3), torch.empty_like(z_inv)) if need_code_grad else (None, None)
        gs, gt = torch.empty_like(s), torch.empty_like(t)
        call(dev, "ls_sdf_backward", self._h, ptr(query), ptr(z_so3), ptr(z_inv), ptr(s), ptr(t), B, M, ptr(sdf), ptr(g), ptr(ws),
                                     ws.numel(), ptr(gq), ptr(gso3), ptr(ginv), ptr(gs), ptr(gt), stream_ptr(dev))
        return gq, gso3, ginv, gs, gt
