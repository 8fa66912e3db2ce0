"""TEST INFRASTRUCTURE ONLY -- canonical k-NN / FPS oracle (numpy + the C twin in ls_oracle.c).

Restates the pytorch3d 0.7.4 ops the reference calls (pytorch3d is pinned in
/root/reference/install.sh:6 but NOT vendored and NOT installed here):

  * ``knn_points``             <- lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141
  * ``sample_farthest_points`` <- vec_dgcnn_atten.py:169, model_utils.py:205, lib_more/more_solver.py:107-108

PARITY UNPINNED for these two ops (no reference golden vector exists):
the arithmetic is *defined* in ls_oracle.c's header; this module is its numpy
twin (contract=0 only: numpy never fuses multiply-add) and the ctypes binding
to the C build (both contract modes).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libls_oracle.so")
_lib = None


def build_c(force=False):
    """Compile ls_oracle.c -> libls_oracle.so with gcc (OpenMP if available)."""
    src = os.path.join(_HERE, "ls_oracle.c")
    if (not force) and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    base = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=c11", src, "-o", _SO, "-lm"]
    try:
        subprocess.check_call(base[:1] + ["-fopenmp"] + base[1:], stderr=subprocess.DEVNULL)
    except Exception:
        subprocess.check_call(base)
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build_c()
        lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        lib.lso_knn.argtypes = [fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, ip, fp]
        lib.lso_knn.restype = None
        lib.lso_fps.argtypes = [fp, ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip]
        lib.lso_fps.restype = None
        _lib = lib
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def knn_c(dst, src, K, contract=0, return_dist=False):
    """dst [B,Nd,3,C], src [B,Ns,3,C] float32 (x-major rows) -> idx [B,Nd,K] int32 (C build)."""
    dst = np.ascontiguousarray(dst, dtype=np.float32)
    src = np.ascontiguousarray(src, dtype=np.float32)
    B, Nd, three, C = dst.shape
    assert three == 3 and src.shape[0] == B and src.shape[2] == 3 and src.shape[3] == C
    Ns = src.shape[1]
    idx = np.empty((B, Nd, K), dtype=np.int32)
    dist = np.empty((B, Nd, K), dtype=np.float32)
    _load().lso_knn(_fp(dst), _fp(src), B, Nd, Ns, C, K, int(contract), _ip(idx), _fp(dist))
    return (idx, dist) if return_dist else idx


def knn_np(dst, src, K):
    """numpy twin of knn_c (contract=0): sequential fp32 sum over j = c*3+x, (dist, idx) order."""
    dst = np.asarray(dst, dtype=np.float32)
    src = np.asarray(src, dtype=np.float32)
    B, Nd, _, C = dst.shape
    Ns = src.shape[1]
    out = np.empty((B, Nd, K), dtype=np.int32)
    for b in range(B):
        d = np.zeros((Nd, Ns), dtype=np.float32)
        for c in range(C):
            for x in range(3):
                diff = dst[b, :, x, c][:, None] - src[b, :, x, c][None, :]
                d = d + diff * diff  # two separately rounded fp32 ops
        order = np.lexsort((np.broadcast_to(np.arange(Ns), d.shape), d), axis=-1)  # (dist, idx)
        out[b] = order[:, :K]
    return out


def fps_c(pts, K, lengths=None, contract=0):
    """pts [B,N,3] float32 -> idx [B,K] int32 (start index 0, first arg-max)."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    B, N, _ = pts.shape
    idx = np.empty((B, K), dtype=np.int32)
    lp = None
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        lp = _ip(lengths)
    _load().lso_fps(_fp(pts), lp, B, N, K, int(contract), _ip(idx))
    return idx


def fps_np(pts, K):
    pts = np.asarray(pts, dtype=np.float32)
    B, N, _ = pts.shape
    out = np.zeros((B, K), dtype=np.int32)
    for b in range(B):
        mind = np.full((N,), np.inf, dtype=np.float32)
        last = 0
        for k in range(1, min(K, N)):
            d = np.zeros((N,), dtype=np.float32)
            for x in range(3):
                diff = pts[b, last, x] - pts[b, :, x]
                d = d + diff * diff
            mind = np.minimum(mind, d)
            last = int(np.argmax(mind))  # first maximum
            out[b, k] = last
    return out
