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
SOURCES = ["model.hip", "knn.hip", "knn_mfma.hip", "knn_xyz.hip", "fps.hip", "gemm.hip", "edge.hip", "edge_fused.hip", "edge_staged.hip", "pointwise.hip", "sdf.hip", "match.hip", "icp.hip", "mise.hip", "mcubes.hip", "sinkhorn.hip", "optim.hip"]
# -fno-slp-vectorize: no COMPILER-FORMED packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  Measured on MI355X (round 2,
# scripts/diag/edge_determinism.py): the attention edge kernel built WITH those instructions was not reproducible while its waves
# shared CUs with the bf16-MFMA GEMM of other streams -- the last 16 lanes of a wave occasionally consumed a stale operand (1e-6..1e-5
# relative on a few points per launch; 4..19 of 48 launches; never when the kernel ran alone); without them 0 of 96, at fewer
# registers (129 -> 119 VGPRs) and the same speed (whole bench 39.7k -> 40.0k obj/s with the flag on every file).  Kernels that
# use packed math on purpose (k-NN distance tiles, the GEMM's bf16 split) spell it with explicit vector types, are unaffected by
# the flag, and are covered by the bit-exactness tests incl. tests/test_hip_fullbatch.py (8 handles in flight).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
         "-ffp-contract=fast-honor-pragmas", "-fno-slp-vectorize"]
# pointwise.hip: the LOOP vectoriser (VF = 2) also forms v_pk_*_f32 (15 in tail_kernel): off for that file
# per-file additions: {"file.hip": [flags]}.  knn_mfma.hip: the 32 x 32 MFMA results of the sweep kernels are consumed element by element by VALU
# compares; in AGPRs (hipcc's default) every element costs a v_accvgpr_read first -- 16 extra VALU instructions per tile in VALU-bound kernels
# fps.hip: LS_PRIO=3 -- the FPS kernels (one workgroup per instance, 512 dependent steps) run with a raised wave priority (ls_common.h: LS_LATENCY_CRITICAL):
# they share their CUs with the chip-filling kernels of layers 0 - 1 and are what layer 2 waits for; one step in flight 40.1k -> 41.2k obj/s, steady state
# unchanged (58.5k / 58.6k); the same priority on the matcher / heads / 32-point k-NN kernels measured neutral and is left off
EXTRA_FLAGS = {"pointwise.hip": ["-fno-vectorize"], "knn_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "fps.hip": ["-DLS_PRIO=3"]}
# Build-time pin of the determinism fix above: the device code of these files is disassembled after every compile and the build FAILS
# if a packed fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) shows up in a kernel whose name contains none of the
# allowed substrings -- a future hipcc, a dropped flag or an innocent float2 cannot silently bring the defect back.  Allowed: the
# kernels that spell packed math on purpose (explicit 2-vectors in the f16 split of the fused attention kernel's staging phase and of
# the weight pre-split), covered by the bit-identity tests with 12 handles in flight (tests/test_hip_fullbatch.py).
PACKED_FP32_GUARD = {"edge.hip": ["edge_attn_fq_kernel", "edge_presplit_wq_kernel"], "edge_fused.hip": ["edge_ft_presplit_w_kernel", "edge_ft_prep_a_kernel"],
                     "edge_staged.hip": ["edge_st_presplit_q_kernel", "edge_attn_staged_kernel"], "pointwise.hip": []}


def llvm_bin():
    """Directory of llvm-objdump: LS_LLVM_BIN, else the toolchain the resolved hipcc belongs to (<rocm>/bin/hipcc -> <rocm>/lib/llvm/bin), else
    /opt/rocm.  A relocated or versioned ROCm (HIPCC=/opt/rocm-x.y/bin/hipcc) is disassembled with ITS objdump."""
    if os.environ.get("LS_LLVM_BIN"):
        return os.environ["LS_LLVM_BIN"]
    import shutil
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hipcc = shutil.which(hipcc) or hipcc
    cand = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "lib", "llvm", "bin")
    return cand if os.path.exists(os.path.join(cand, "llvm-objdump")) else "/opt/rocm/lib/llvm/bin"


# host-only translation units (no device code): g++, strict fp (no contraction: simplify.cpp reproduces the reference's doubles bit for bit)
HOST_SOURCES = ["simplify.cpp"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off"]


def _newest_src():
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def packed_fp32_report(obj):
    """{kernel symbol: count of v_pk_(mul|add|fma)_f32} in the gfx950 code object embedded in the host object `obj`."""
    import re
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="ls_pkchk_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        objdump = os.path.join(llvm_bin(), "llvm-objdump")
        if not os.path.exists(objdump):
            raise RuntimeError(f"packed-fp32 guard: {objdump} not found (set LS_LLVM_BIN to the directory of the llvm-objdump that belongs to "
                               f"your hipcc; the guard is part of the build -- DESIGN.md, determinism)")
        subprocess.run([objdump, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not dev:
            raise RuntimeError(f"packed-fp32 guard: no gfx950 bundle found in {obj}")
        dis = subprocess.run([objdump, "-d", os.path.join(tmp, dev[0])], check=True, capture_output=True,
                             text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    counts, name = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            name = m.group(1)
        elif name and re.search(r"\bv_pk_(mul|add|fma)_f32\b", line):
            counts[name] = counts.get(name, 0) + 1
    return counts


def check_packed_fp32(objdir):
    bad = []
    for src, allowed in PACKED_FP32_GUARD.items():
        for kern, n in packed_fp32_report(os.path.join(objdir, src.replace(".hip", ".o"))).items():
            if not any(a in kern for a in allowed):
                bad.append(f"{src}: {kern}: {n} packed fp32 instruction(s)")
    if bad:
        raise RuntimeError("packed fp32 math (v_pk_*_f32) in kernels that must not have it (build.py PACKED_FP32_GUARD; was "
                           "-fno-slp-vectorize / -fno-vectorize dropped?):\n  " + "\n  ".join(bad))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_src():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = {k: list(v) for k, v in EXTRA_FLAGS.items()}
    if any("-amdgpu-mfma-vgpr-form" in f for v in extra.values() for f in v):
        # an LLVM-internal option (a pure instruction-count optimisation, the kernels are correct in either register form): dropped when this hipcc lacks it
        probe = subprocess.run([hipcc, "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-x", "hip", "-c", "-o", os.devnull, "-"],
                               input=b"__global__ void ls_probe() {}\n", stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if probe.returncode != 0:
            extra = {k: [f for f in v if f != "-mllvm" and "-amdgpu-mfma-vgpr-form" not in f] for k, v in extra.items()}
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    hdr_t = max(hdr_t, os.path.getmtime(os.path.join(HERE, "..", "include", "livingscenes_hip.h")))
    procs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            continue
        cmd = [hipcc] + FLAGS + extra.get(s, []) + ["-x", "hip", "-c", src, "-o", obj]
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
    check_packed_fp32(objdir)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
