"""Build liblivingscenes_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m livingscenes_amd.build [--force]

The library is built IN-TREE (livingscenes_amd/lib/) so that it travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblivingscenes_hip.so")
SOURCES = ["model.hip", "knn.hip", "knn_mfma.hip", "knn_xyz.hip", "fps.hip", "gemm.hip", "edge.hip", "pointwise.hip", "sdf.hip", "match.hip", "icp.hip", "mise.hip", "mcubes.hip", "sinkhorn.hip", "optim.hip"]
# -fno-slp-vectorize: no COMPILER-FORMED packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  Measured on MI355X (round 2,
# scripts/diag/edge_determinism.py): the attention edge kernel built WITH those instructions was not reproducible while its waves
# shared CUs with the bf16-MFMA GEMM of other streams -- the last 16 lanes of a wave occasionally consumed a stale operand (1e-6..1e-5
# relative on a few points per launch; 4..19 of 48 launches; never when the kernel ran alone); without them 0 of 96, at fewer
# registers (129 -> 119 VGPRs) and the same speed (whole bench 39.7k -> 40.0k obj/s with the flag on every file).  Kernels that
# use packed math on purpose (k-NN distance tiles, the GEMM's bf16 split) spell it with explicit vector types, are unaffected by
# the flag, and are covered by the bit-exactness tests incl. tests/test_hip_fullbatch.py (8 handles in flight).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
         "-ffp-contract=fast-honor-pragmas", "-fno-slp-vectorize"]
EXTRA_FLAGS = {}   # per-file additions: {"file.hip": [flags]}
# host-only translation units (no device code): g++, strict fp (no contraction: simplify.cpp reproduces the reference's doubles bit for bit)
HOST_SOURCES = ["simplify.cpp"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def _newest_src():
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_src():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_t = max(hdr_t, os.path.getmtime(os.path.join(HERE, "..", "include", "livingscenes_hip.h")))
    procs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s in HOST_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".cpp", ".o"))
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [os.environ.get("CXX", "g++")] + HOST_FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append((s, out.decode(errors="replace")))
        elif verbose and out:
            print(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {s}\n{o}" for s, o in failed))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
