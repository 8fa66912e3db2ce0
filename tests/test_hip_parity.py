"""GPU parity tests: every HIP operator (called through the C ABI via ctypes) against the CPU oracle on the same
seeded inputs, and against the golden fixtures generated from the imported reference.

Bars (BASELINE.json north_star): integer outputs (k-NN / FPS indices, match assignments) BIT-EXACT on identical
inputs; fp32 outputs within 1e-4 of the tensor max-norm (TOL below; most are ~1e-6)."""
import os

import numpy as np
import pytest
import torch

from livingscenes_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4  # relative to the max-norm of the reference tensor (north_star: "fp32 SDF/pose within 1e-4 rel")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def relerr(a, b):
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rows(f):
    """reference layout [B,C,3,N] -> library layout [B,N,3,C]"""
    return f.permute(0, 3, 2, 1).contiguous()


# ------------------------------------------------------------------------------------------------ k-NN
@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("shape", [(2, 100, 150, 1), (2, 70, 130, 32), (1, 64, 64, 64), (3, 33, 257, 96), (1, 5, 9, 32),
                                   (2, 32, 128, 128), (2, 32, 32, 256), (1, 20, 45, 160), (2, 9, 70, 64)])
def test_knn_bit_exact(shape, contract):
    from livingscenes_amd import ops
    from oracle import canon
    B, Nd, Ns, C = shape
    rng = np.random.default_rng(hash(shape) % 2**31)
    src = rng.standard_normal((B, Ns, 3, C)).astype(np.float32)
    dst = rng.standard_normal((B, Nd, 3, C)).astype(np.float32)
    src[0, 7] = src[0, 2]        # exact duplicate candidates -> distance ties (lower index wins)
    dst[0, 0] = src[0, 2]        # zero distance
    ref, refd = canon.knn_c(dst, src, 16, contract=contract, return_dist=True)
    idx, dist = ops.knn(torch.from_numpy(dst).to(_dev()), torch.from_numpy(src).to(_dev()), 16, flags=contract, return_dist=True)
    assert np.array_equal(idx.cpu().numpy(), ref)
    valid = ref >= 0
    assert np.array_equal(dist.cpu().numpy()[valid], refd[valid])  # distances bit-exact too


@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("case", ["tiny", "one_chunk", "multi_chunk", "ties", "rows", "few"])
def test_knn_xyz_wave_kernel_bit_exact(case, contract, monkeypatch):
    """Raw-cloud (C == 1) k-NN: wave-per-query kernel incl. fewer candidates than K, several 1024-candidate chunks, more exact
    ties than one sorting round holds, query rows."""
    from livingscenes_amd import ops
    from oracle import canon
    rng = np.random.default_rng(11)
    B, Nd, Ns = {"tiny": (3, 7, 9), "one_chunk": (2, 300, 1024), "multi_chunk": (2, 130, 2500), "ties": (1, 40, 700),
                 "rows": (2, 50, 1024), "few": (2, 100, 150)}[case]
    src = rng.standard_normal((B, Ns, 3, 1)).astype(np.float32)
    dst = rng.standard_normal((B, Nd, 3, 1)).astype(np.float32)
    dst_rows = None
    if case == "ties":        # 200 copies of 3 distinct points: hundreds of candidates tie at the K-th distance
        src[0, :600] = src[0, :3].repeat(200, axis=0)
        dst[0, 0] = src[0, 1]
    if case == "rows":
        dst_rows = np.stack([rng.permutation(Ns)[:Nd] for _ in range(B)]).astype(np.int32)
        ref, refd = canon.knn_c(np.stack([src[b][dst_rows[b]] for b in range(B)]), src, 16, contract=contract, return_dist=True)
        dst_t = torch.from_numpy(src).to(_dev())
    else:
        ref, refd = canon.knn_c(dst, src, 16, contract=contract, return_dist=True)
        dst_t = torch.from_numpy(dst).to(_dev())
    idx, dist = ops.knn(dst_t, torch.from_numpy(src).to(_dev()), 16, flags=contract, return_dist=True,
                        dst_rows=None if dst_rows is None else torch.from_numpy(dst_rows).to(_dev()))
    assert np.array_equal(idx.cpu().numpy(), ref)
    valid = ref >= 0
    assert np.array_equal(dist.cpu().numpy()[valid], refd[valid])


@pytest.mark.parametrize("K", [1, 5, 16])
@pytest.mark.parametrize("shape", [(2, 90, 300, 1), (2, 90, 300, 32), (2, 90, 300, 64), (2, 20, 300, 128)])
def test_knn_smaller_k_on_every_kernel(shape, K):
    """K < 16 through the raw-cloud kernel, the seeded sweep (C = 32 / 64: threshold = the K-th hint), the small-problem kernel."""
    from livingscenes_amd import ops
    from oracle import canon
    B, Nd, Ns, C = shape
    rng = np.random.default_rng(K * 100 + C)
    src = rng.standard_normal((B, Ns, 3, C)).astype(np.float32)
    dst = rng.standard_normal((B, Nd, 3, C)).astype(np.float32)
    ref, refd = canon.knn_c(dst, src, K, return_dist=True)
    seeds = None
    if C in (32, 64):
        full = canon.knn_c(dst, src, 16)
        seeds = torch.from_numpy(np.where(rng.random(full.shape) < 0.7, full, rng.integers(0, Ns, full.shape)).astype(np.int32)).to(_dev())
    idx, dist = ops.knn(torch.from_numpy(dst).to(_dev()), torch.from_numpy(src).to(_dev()), K, return_dist=True, seeds=seeds)
    assert np.array_equal(idx.cpu().numpy(), ref) and np.array_equal(dist.cpu().numpy(), refd)


def test_knn_large_batch_unsplit_path():
    """B*qtiles >= 768 workgroups -> the single-pass (no candidate split) path, at the encoder's layer-1 shape."""
    from livingscenes_amd import ops
    from oracle import canon
    rng = np.random.default_rng(77)
    from livingscenes_amd import _lib
    f = rng.standard_normal((48, 1024, 3, 32)).astype(np.float32)
    ref = canon.knn_c(f, f, 16)
    ft = torch.from_numpy(f).to(_dev())
    assert np.array_equal(ops.knn(ft, ft, 16).cpu().numpy(), ref)
    # seeded launches take the MFMA sweep kernel (knn_mfma.hip) unless FLAG_KNN_VALU_ONLY forces the all-VALU kernel:
    # hints = the lists of a perturbed copy of the features (what the previous layer's graph is to the next layer)
    hints = torch.from_numpy(canon.knn_c((f + 0.3 * rng.standard_normal(f.shape)).astype(np.float32), f, 16)).to(_dev())
    assert np.array_equal(ops.knn(ft, ft, 16, seeds=hints).cpu().numpy(), ref)
    assert np.array_equal(ops.knn(ft, ft, 16, seeds=hints, flags=_lib.FLAG_KNN_VALU_ONLY).cpu().numpy(), ref)


@pytest.mark.parametrize("N", [640, 480])     # 480: un-seeded calls take the one-sweep path (<= 512 candidates), 640 the class-winner hints
@pytest.mark.parametrize("C", [32, 64])
@pytest.mark.parametrize("case", ["offset", "near_duplicates", "clustered", "scale_mix"])
def test_knn_mfma_filter_is_exact_on_adversarial_features(case, C, N):
    """The MFMA sweep kernel may only drop pairs that provably cannot enter a list.  Stress the cancellation in
    |q|^2+|s|^2-2q.s: a huge common offset, near-duplicate points, tight clusters, wildly different norms.  The result
    must be bit-identical to the oracle for the all-VALU kernel (un-seeded) AND the seeded MFMA sweep kernel."""
    from livingscenes_amd import _lib, ops
    from oracle import canon
    rng = np.random.default_rng({"offset": 1, "near_duplicates": 2, "clustered": 3, "scale_mix": 4}[case])
    B = 3
    f = rng.standard_normal((B, N, 3, C)).astype(np.float32)
    if case == "offset":
        f = (f * 1e-3 + 50.0).astype(np.float32)            # norms ~ 7e5, spreads ~ 1e-3: d^ is pure cancellation noise
    elif case == "near_duplicates":
        f[:, 1::2] = f[:, 0::2] + (rng.standard_normal((B, N // 2, 3, C)) * 1e-6).astype(np.float32)
        f[:, 10] = f[:, 11]                                   # exact duplicates too
    elif case == "clustered":
        cen = rng.standard_normal((B, 8, 3, C)).astype(np.float32) * 5
        f = (cen[:, rng.integers(0, 8, N)] + f * 1e-2).astype(np.float32)
    else:
        f = (f * np.exp(rng.uniform(-6, 6, (B, N, 1, 1)))).astype(np.float32)
    ref, refd = canon.knn_c(f, f, 16, return_dist=True)
    ft = torch.from_numpy(f).to(_dev())
    idx, dist = ops.knn(ft, ft, 16, return_dist=True)
    assert np.array_equal(idx.cpu().numpy(), ref) and np.array_equal(dist.cpu().numpy(), refd)
    # seeded sweep kernel (thresholds from hints): exact hints, hints shifted to other points' lists, half-garbage hints
    hints = {"exact": ref, "shifted": np.roll(ref, 7, axis=1),
             "mixed": np.where(rng.random(ref.shape) < 0.5, ref, rng.integers(0, N, ref.shape)).astype(np.int32)}
    for name, h in hints.items():
        for fl in (0, _lib.FLAG_CONTRACT_FMA):
            r2, d2 = (ref, refd) if not (fl & _lib.FLAG_CONTRACT_FMA) else canon.knn_c(f, f, 16, contract=1, return_dist=True)
            i3, d3 = ops.knn(ft, ft, 16, flags=fl, seeds=torch.from_numpy(np.ascontiguousarray(h)).to(_dev()), return_dist=True)
            assert np.array_equal(i3.cpu().numpy(), r2) and np.array_equal(d3.cpu().numpy(), d2), (name, fl)


@pytest.mark.parametrize("shape", [(2, 200, 200, 32), (3, 70, 330, 64), (40, 256, 1024, 32), (3, 70, 330, 32), (1, 33, 100, 32),
                                   (8, 128, 512, 64), (1, 64, 2500, 32), (1, 40, 2100, 64),   # Ns > 2048: fp32 sweep kernel
                                   (1, 70, 2048, 32), (1, 50, 1500, 64)])                      # largest bf16-sweep bitmaps
def test_knn_hints_do_not_change_the_result(shape):
    """seed_idx are HINTS: exact neighbours, random indices, duplicates of each other's ranges, -1 and out-of-range values
    must all yield the oracle's answer bit for bit (also through the candidate-split + merge path)."""
    from livingscenes_amd import ops
    from oracle import canon
    B, Nd, Ns, C = shape
    rng = np.random.default_rng(B * 1000 + Nd)
    src = rng.standard_normal((B, Ns, 3, C)).astype(np.float32)
    sel = np.stack([rng.permutation(Ns)[:Nd] for _ in range(B)]).astype(np.int32)
    dst = np.stack([src[b, sel[b]] for b in range(B)])
    ref, refd = canon.knn_c(dst, src, 16, return_dist=True)
    st, dt = torch.from_numpy(src).to(_dev()), torch.from_numpy(dst).to(_dev())
    hints = {
        "exact": ref.copy(),
        "random": rng.integers(0, Ns, (B, Nd, 16)).astype(np.int32),
        "mixed": np.where(rng.random((B, Nd, 16)) < 0.5, ref, rng.integers(-3, Ns + 5, (B, Nd, 16))).astype(np.int32),
        "none": -np.ones((B, Nd, 16), np.int32),
    }
    hints["random"][:, :, 1] = hints["random"][:, :, 0]  # duplicate hints inside a row
    from livingscenes_amd import _lib
    for name, h in hints.items():
        for fl in (0, _lib.FLAG_KNN_VALU_ONLY):   # MFMA sweep kernel (C == 32, 64) / all-VALU kernel
            idx, dist = ops.knn(dt, st, 16, seeds=torch.from_numpy(h).to(_dev()), return_dist=True, flags=fl)
            assert np.array_equal(idx.cpu().numpy(), ref), (name, fl)
            assert np.array_equal(dist.cpu().numpy(), refd), (name, fl)


def test_knn_dst_rows_and_self():
    from livingscenes_amd import ops
    from oracle import canon
    rng = np.random.default_rng(5)
    f = rng.standard_normal((2, 300, 3, 32)).astype(np.float32)
    sel = np.stack([rng.permutation(300)[:90] for _ in range(2)]).astype(np.int32)
    ref = canon.knn_c(np.stack([f[b, sel[b]] for b in range(2)]), f, 16)
    ft = torch.from_numpy(f).to(_dev())
    idx = ops.knn(ft, ft, 16, dst_rows=torch.from_numpy(sel).to(_dev()))
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert (idx[:, :, 0].cpu().numpy() == sel).all()  # a point is its own nearest neighbour


# ------------------------------------------------------------------------------------------------ FPS
@pytest.mark.parametrize("contract", [0, 1])
@pytest.mark.parametrize("N,K", [(128, 32), (200, 70), (256, 64), (512, 128), (1024, 512), (1500, 300), (2048, 600), (5000, 1024), (20000, 1024)])
def test_fps_bit_exact(N, K, contract):
    from livingscenes_amd import ops
    from oracle import canon
    rng = np.random.default_rng(N + K)
    p = rng.standard_normal((3, N, 3)).astype(np.float32)
    p[1, 10] = p[1, 3]
    ref = canon.fps_c(p, K, contract=contract)
    idx, pts = ops.fps(torch.from_numpy(p).to(_dev()), K, flags=contract, return_points=True)
    assert np.array_equal(idx.cpu().numpy(), ref)
    assert np.array_equal(pts.cpu().numpy(), np.take_along_axis(p, ref[..., None].astype(np.int64), 1))


@pytest.mark.parametrize("contract", [0, 1])
def test_fps_raw_cloud_bucketed_scan_bit_exact(contract):
    """Clouds above 8 192 points take the bucketed, pruned scan (fps.hip: fps_bucket_kernel): same indices as the full scan of the
    oracle on scanner-like clouds (surfaces, not blobs), ragged lengths on both sides of the 8^3 / 16^3 grid switch, duplicated
    points (tie -> smallest original index), a planar and a fully degenerate cloud."""
    from livingscenes_amd import ops
    from oracle import canon
    rng = np.random.default_rng(77 + contract)
    N = 60000
    p = np.zeros((7, N, 3), np.float32)
    u = rng.random((N, 2)).astype(np.float32)
    p[0] = np.stack([u[:, 0] * 3, u[:, 1] * 2, 0.2 * np.sin(4 * u[:, 0]) + 0.01 * rng.standard_normal(N)], 1)    # a bumpy sheet
    d = rng.standard_normal((N, 3)).astype(np.float32)
    p[1] = d / np.linalg.norm(d, axis=1, keepdims=True) * np.float32(0.7) + np.float32(5.0)                         # a sphere shell, off-centre
    p[2] = rng.standard_normal((N, 3)).astype(np.float32)
    p[2, 30000:] = p[2, :30000]                                                                                   # every point twice
    p[3] = rng.standard_normal((N, 3)).astype(np.float32); p[3, :, 2] = 1.5                                        # planar: one axis has no extent
    p[4] = 0.25                                                                                                   # all points identical
    p[5] = rng.standard_normal((N, 3)).astype(np.float32) * np.float32(1e-3)
    p[6] = rng.standard_normal((N, 3)).astype(np.float32)
    lens = np.array([60000, 60000, 60000, 20000, 9000, 12000, 700], np.int32)
    ref = canon.fps_c(p, 1024, lengths=lens, contract=contract)
    idx, pts = ops.fps(torch.from_numpy(p).to(_dev()), 1024, lengths=torch.from_numpy(lens), flags=contract, return_points=True)
    assert np.array_equal(idx.cpu().numpy(), ref)
    sel = np.take_along_axis(p, np.maximum(ref, 0)[..., None].astype(np.int64), 1) * (ref >= 0)[..., None]
    assert np.array_equal(pts.cpu().numpy(), sel)
    q = rng.standard_normal((2, 8200, 3)).astype(np.float32)                                                       # just above the switch, K > n on one
    lq = np.array([8200, 100], np.int32)
    assert np.array_equal(ops.fps(torch.from_numpy(q).to(_dev()), 300, lengths=torch.from_numpy(lq)).cpu().numpy(),
                          canon.fps_c(q, 300, lengths=lq))


def test_fps_ragged_and_degenerate():
    from livingscenes_amd import ops
    from oracle import canon
    rng = np.random.default_rng(2)
    p = rng.standard_normal((3, 700, 3)).astype(np.float32)
    lens = np.array([700, 50, 3], np.int32)
    ref = canon.fps_c(p, 64, lengths=lens)
    idx = ops.fps(torch.from_numpy(p).to(_dev()), 64, lengths=torch.from_numpy(lens))
    assert np.array_equal(idx.cpu().numpy(), ref)
    z = np.zeros((1, 40, 3), np.float32)  # all-duplicate cloud
    assert np.array_equal(ops.fps(torch.from_numpy(z).to(_dev()), 8).cpu().numpy(), canon.fps_c(z, 8))


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (300, 200, 64), (1000, 640, 128), (77, 257, 32), (192, 2048, 512), (5, 1, 4)])
def test_gemm(M, N, K):
    from livingscenes_amd import ops
    g = torch.Generator().manual_seed(M * N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ref = A.double() @ W.double().T + b.double()
    out = ops.gemm(A.to(_dev()), W.to(_dev()), b.to(_dev()))
    assert relerr(out, ref) < 2e-6
    out = ops.gemm(A.to(_dev()), W.to(_dev()), None, relu=True)
    assert relerr(out, (A.double() @ W.double().T).clamp(min=0)) < 2e-6
    # MFMA f32 == fmaf chain: an asymmetric integer-valued problem must be EXACT (also catches transposes)
    Ai = torch.randint(-4, 5, (M, K), generator=g).float()
    Wi = torch.randint(-4, 5, (N, K), generator=g).float() + torch.arange(N)[:, None].float()
    assert torch.equal(ops.gemm(Ai.to(_dev()), Wi.to(_dev())).cpu(), Ai @ Wi.T)


def test_gemm_split_is_as_accurate_as_the_fp32_chain_and_row_invariant():
    """The default GEMM forms every fp32 product from two f16 pieces on the f16 matrix cores (gemm.hip;
    LS_GEMM_MODE=bf16x3: three bf16 pieces -- test_gemm_mode_switches).  Componentwise error against
    fp64 must stay within the fp32 FMA chain's own bound (in units of 2^-24 sum|a||w|: measured 6-9 vs 11-13 for the chain) on
    badly scaled operands, and the kernel's arithmetic must not depend on M: a row's result is the same whatever other rows the call
    carries (checked here where ls_gemm_f32 does not split K, K < 128; the decoder path, which never splits K, is covered by
    test_ragged_decode_and_batched_mise_equal_per_instance)."""
    from livingscenes_amd import ops
    if os.environ.get("LS_GEMM_MODE") == "fp32":
        pytest.skip("the fp32-MFMA mode is an fp32 FMA chain itself (11 - 16 units on these operands)")
    for K in (32, 64, 256, 768):
        g = torch.Generator().manual_seed(K)
        A = torch.randn(3000, K, generator=g) * torch.exp(2 * torch.randn(3000, K, generator=g))
        W = torch.randn(260, K, generator=g) * torch.exp(2 * torch.randn(260, K, generator=g))
        out = ops.gemm(A.to(_dev()), W.to(_dev())).cpu().double()
        ref = A.double() @ W.double().T
        unit = (A.double().abs() @ W.double().abs().T) * 2.0 ** -24
        assert ((out - ref).abs() / unit).max() < 16.0, K
        if K < 128:
            part = ops.gemm(A[100:137].contiguous().to(_dev()), W.to(_dev())).cpu()
            assert torch.equal(part.double(), out[100:137]), K


def test_gemm_mode_switches():
    """The process-wide arithmetic modes (gemm.hip: gemm_mode, read once from LS_GEMM_MODE): bf16x3 -- six-MFMA split, any fp32 range, here with
    operands far outside the f16 range -- and fp32 -- exact fp32 FMA chains on v_mfma_f32_32x32x2_f32 -- keep the tolerance."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch; from livingscenes_amd import ops; g = torch.Generator().manual_seed(1);"
            "A = torch.randn(700, 256, generator=g) * 3e6; W = torch.randn(200, 256, generator=g) * 1e-7;"
            "o = ops.gemm(A.cuda(), W.cuda()).cpu().double(); r = A.double() @ W.double().T;"
            "assert torch.isfinite(o).all() and ((o - r).abs().max() / r.abs().max()) < 2e-6")
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, LS_GEMM_MODE="bf16x3"), cwd=root)
    code = ("import torch; from livingscenes_amd import ops; g = torch.Generator().manual_seed(1);"
            "A = torch.randn(700, 96, generator=g); W = torch.randn(200, 96, generator=g);"
            "o = ops.gemm(A.cuda(), W.cuda()).cpu().double(); r = A.double() @ W.double().T;"
            "assert ((o - r).abs().max() / r.abs().max()) < 2e-6")
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, LS_GEMM_MODE="fp32"), cwd=root)


# ------------------------------------------------------------------------------------------------ prologue
def test_prologue():
    from livingscenes_amd import ops
    x = synth.make_instances(4, 1024, seed=11)
    pts, cen, sc = ops.encode_prologue(x.to(_dev()))
    c = x.mean(-1)
    xc = x - c[..., None]
    d = torch.cdist(xc.transpose(1, 2).double(), xc.transpose(1, 2).double())
    s0 = d.view(4, -1).topk(5, dim=-1)[0].mean(-1)
    assert relerr(cen, c) < 1e-6 and relerr(sc, s0) < 1e-6
    assert relerr(pts, (xc / s0[:, None, None].float()).transpose(1, 2)) < 2e-6


@pytest.mark.parametrize("case", ["sphere", "tiny", "duplicates", "line", "ragged_n"])
def test_prologue_pruned_pair_scan_edge_cases(case):
    """The prologue prunes the N^2 pair scan to hull-side points; the pruning must be exact: points on a sphere (nothing can
    be pruned), N = 3 / 4 (fewer than three partners), exact duplicates of the extreme points, a collinear cloud."""
    from livingscenes_amd import ops
    g = torch.Generator().manual_seed(5)
    if case == "sphere":
        x = torch.randn(3, 3, 700, generator=g); x = x / x.norm(dim=1, keepdim=True) + torch.tensor([0.3, -0.2, 0.1])[None, :, None]
    elif case == "tiny":
        x = torch.randn(2, 3, 3, generator=g)
    elif case == "duplicates":
        x = torch.randn(2, 3, 256, generator=g); x[:, :, 1] = x[:, :, 0] = 4.0; x[:, :, 5] = x[:, :, 4] = -4.0
    elif case == "line":
        t = torch.linspace(-1, 1, 333)[None, None, :]; x = torch.tensor([1.0, 2.0, -0.5])[None, :, None] * t + 0.25
        x = x.repeat(2, 1, 1).contiguous()
    else:
        x = torch.randn(2, 3, 1001, generator=g) * torch.tensor([3.0, 1.0, 0.3])[None, :, None]
    B = x.shape[0]
    pts, cen, sc = ops.encode_prologue(x.to(_dev()))
    c = x.mean(-1)
    xc = x - c[..., None]
    d = torch.cdist(xc.transpose(1, 2).double(), xc.transpose(1, 2).double())
    s0 = d.view(B, -1).topk(5, dim=-1)[0].mean(-1)
    assert relerr(cen, c) < 1e-6 and relerr(sc, s0) < 2e-6
    assert relerr(pts, (xc / s0[:, None, None].float()).transpose(1, 2)) < 4e-6


# ------------------------------------------------------------------------------------------------ encoder
def _hip_model(ecfg, ew, dcfg=None, dw=None):
    from livingscenes_amd import ops, packing
    desc, blob = packing.pack_model(ew, ecfg, dw, dcfg)
    return ops.HipModel(desc, blob, _dev())


@pytest.mark.parametrize("which", ["small", "full"])
def test_encoder_layerwise_knn_on_oracle_inputs(which):
    """Bit-exact k-NN / FPS at the encoder's real shapes: feed the ORACLE's per-layer features to the HIP k-NN."""
    from livingscenes_amd import ops
    from oracle import net
    if which == "small":
        cfg, B, N, seed = synth.small_encoder_cfg(), 2, 128, 7
    else:
        cfg, B, N, seed = synth.default_encoder_cfg(), 2, 1024, 0
    w = synth.make_encoder_weights(cfg, seed)
    x = synth.make_instances(B, N, seed=3, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.3
    tr = {}
    net.encoder_forward(w, cfg, x, trace=tr)
    for i in range(cfg["num_layers"]):
        src, dst = rows(tr[f"src_f_{i}"]), rows(tr[f"dst_f_in_{i}"])
        idx = ops.knn(dst.to(_dev()), src.to(_dev()), 16)
        assert np.array_equal(idx.cpu().numpy(), tr[f"knn_idx_{i}"].numpy().astype(np.int32)), f"layer {i}"


@pytest.mark.parametrize("which,B,N", [("small", 2, 128), ("small", 5, 256), ("small", 1, 200), ("small", 3, 333),
                                       ("full", 2, 1024), ("full", 3, 1024), ("full", 1, 1000)])
def test_encoder_forward_vs_oracle(which, B, N):
    """VecDGCNN_att.forward (pre-normalised input): FPS and layer-0 k-NN indices bit-exact (identical inputs);
    deeper k-NN layers search in features that differ by fp32 round-off, so they are compared as a match rate;
    codes within TOL of max-norm.  B=3 pins the torch.cross decision (always xyz)."""
    from oracle import net
    if which == "small":
        cfg, seed = synth.small_encoder_cfg(), 7
    else:
        cfg, seed = synth.default_encoder_cfg(), 0
    w = synth.make_encoder_weights(cfg, seed)
    x = synth.make_instances(B, N, seed=5, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.1
    tr = {}
    center, scale, z_so3, z_inv = net.encoder_forward(w, cfg, x, trace=tr)
    m = _hip_model(cfg, w)
    hz, hi, hs, ht, knn_l, fps_l = m.encode(x.to(_dev()), pre_normalised=True, trace=True)
    assert np.array_equal(knn_l[0].cpu().numpy(), tr["knn_idx_0"].numpy().astype(np.int32))
    for j, i in enumerate(cfg["down_sample_layers"]):
        assert np.array_equal(fps_l[j].cpu().numpy(), tr[f"fps_idx_{i}"].numpy().astype(np.int32)), f"fps level {j}"
    flipped = False
    for i in range(1, cfg["num_layers"]):
        a, b = knn_l[i].cpu().numpy(), tr[f"knn_idx_{i}"].numpy()
        rate = (a == b).mean()
        assert rate > 0.995, f"layer {i}: k-NN index agreement {rate:.5f}"
        flipped |= rate < 1.0
    if flipped:
        # a feature-space near-tie went the other way somewhere (see oracle.net.encoder_forward): every differing entry must be a
        # near-tie in the ORACLE's own features, and on the device's graph the oracle must reproduce the device's codes
        tr2 = {}
        graph = {i: knn_l[i].cpu() for i in range(1, cfg["num_layers"])}
        center, scale, z_so3, z_inv = net.encoder_forward(w, cfg, x, trace=tr2, graph=graph)
        for i in range(1, cfg["num_layers"]):
            a, b = knn_l[i].cpu().numpy().astype(np.int64), tr2[f"knn_idx_{i}"].numpy()
            assert np.array_equal(a, b)
            ref_idx = tr[f"knn_idx_{i}"].numpy()
            if np.array_equal(a, ref_idx) or not torch.equal(tr[f"src_f_{i}"], tr2[f"src_f_{i}"]):
                continue   # (below the first flipped layer the two runs see different features: only the first one can be judged)
            src = tr[f"src_f_{i}"].reshape(B, -1, tr[f"src_f_{i}"].shape[-1]).transpose(1, 2).double().numpy()
            dst = tr[f"dst_f_in_{i}"].reshape(B, -1, tr[f"dst_f_in_{i}"].shape[-1]).transpose(1, 2).double().numpy()
            bb, nn, kk = np.nonzero(a != ref_idx)
            d_dev = ((dst[bb, nn] - src[bb, a[bb, nn, kk]]) ** 2).sum(-1)
            d_ref = ((dst[bb, nn] - src[bb, ref_idx[bb, nn, kk]]) ** 2).sum(-1)
            assert (np.abs(d_dev - d_ref) <= 1e-5 * np.maximum(d_ref, 1e-30)).all(), f"layer {i}: a flipped neighbour is not a near-tie"
    assert relerr(hz, z_so3) < TOL and relerr(hi, z_inv) < TOL
    assert relerr(hs, scale) < TOL and relerr(ht, center.squeeze(1)) < TOL


def test_encoder_b3_equals_the_reference_per_instance(golden):
    """The B = 3 decision pinned by the REFERENCE (tests/golden/encoder_b3.npz, make_golden_b3.py): ls_encode on a batch of exactly three instances
    equals the reference's VecDGCNN_att.forward run on the three instances one at a time (vec_dgcnn_atten.py:157 crosses over the batch axis at
    B = 3; the build never does): FPS and layer-0 k-NN identical, codes within TOL."""
    g = golden("encoder_b3")
    cfg = synth.small_encoder_cfg()
    m = _hip_model(cfg, synth.make_encoder_weights(cfg, 7))
    hz, hi, hs, ht, knn_l, fps_l = m.encode(torch.from_numpy(g["x"]).to(_dev()), pre_normalised=True, trace=True)
    for b in range(3):
        assert np.array_equal(knn_l[0][b:b + 1].cpu().numpy(), g[f"single{b}_knn_idx_0"])
        assert np.array_equal(fps_l[0][b:b + 1].cpu().numpy(), g[f"single{b}_fps_idx_0"])
        for i in range(1, cfg["num_layers"]):
            assert (knn_l[i][b:b + 1].cpu().numpy() == g[f"single{b}_knn_idx_{i}"]).mean() > 0.995
    assert relerr(hz, g["single_z_so3"]) < TOL and relerr(hi, g["single_z_inv"]) < TOL
    assert relerr(hs, g["single_scale"]) < TOL and relerr(ht, g["single_center"].reshape(3, 3)) < TOL
    assert relerr(hz, g["batched_z_so3"]) > 0.1          # ... and not the reference's batched B = 3 result


def test_encoder_with_more_than_8192_points_uses_the_planned_fps_scratch():
    """the encoder's own FPS chain passed no workspace, so a cloud of 8 193 .. 65 536 points (fps.hip: exact bucket pruning, which needs
    scratch) failed in the middle of the enqueue.  The plan reserves the scratch now: FPS and layer-0 k-NN indices bit-exact, codes within TOL."""
    from oracle import net
    cfg = synth.small_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 7)
    x = synth.make_instances(1, 9216, seed=5, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.1
    tr = {}
    center, scale, z_so3, z_inv = net.encoder_forward(w, cfg, x, trace=tr)
    m = _hip_model(cfg, w)
    hz, hi, hs, ht, knn_l, fps_l = m.encode(x.to(_dev()), pre_normalised=True, trace=True)
    assert np.array_equal(fps_l[0].cpu().numpy(), tr["fps_idx_2"].numpy().astype(np.int32))
    assert np.array_equal(knn_l[0].cpu().numpy(), tr["knn_idx_0"].numpy().astype(np.int32))
    assert (knn_l[1].cpu().numpy() == tr["knn_idx_1"].numpy()).mean() > 0.995
    assert relerr(hz, z_so3) < TOL and relerr(hi, z_inv) < TOL and relerr(hs, scale) < TOL and relerr(ht, center.reshape(1, 3)) < TOL
    # more points than FPS handles: refused BEFORE anything is enqueued (workspace query answers 0, encode raises)
    with pytest.raises(Exception, match="65536"):
        m.encode(torch.zeros(1, 3, 70000, device=_dev()), pre_normalised=True)


def test_shape_prior_encode_vs_golden(golden):
    """Shape_Prior.encode end to end against the fixture produced by the reference's own model_utils.Shape_Prior."""
    g = golden("shape_prior_full")
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    m = _hip_model(ecfg, ew, dcfg, dw)
    x = synth.make_instances(2, 1024, seed=0)
    z_so3, z_inv, s, t, knn_l, fps_l = m.encode(x.to(_dev()), trace=True)
    for k, v in (("z_so3", z_so3), ("z_inv", z_inv), ("s", s), ("t", t.unsqueeze(1))):
        assert relerr(v, g[k]) < TOL, k
    for j in range(3):
        assert np.array_equal(fps_l[j].cpu().numpy(), g[f"fps_idx_{j}"].astype(np.int32))
    # SDF queries with the golden code
    code = {k: torch.from_numpy(g[k]).to(_dev()) for k in ("z_so3", "z_inv", "s", "t")}
    sdf = m.sdf_decode(torch.from_numpy(g["query"]).to(_dev()), code["z_so3"], code["z_inv"], code["s"], code["t"])
    assert relerr(sdf, g["sdf"]) < TOL


def test_load_ckpt_from_log_on_device_vs_golden(golden, tmp_path, monkeypatch):
    """the reference's LOADER entry on the device -- eval_3rscan.py:505-510 / eval_flyingshape.py:182 call
    `load_ckpt_from_log(ckpt_dir)` (model_utils.py:267-283: CWD-relative ./configs/room4cates.yaml, exactly one <ckpt>/checkpoint/*latest.pt and
    one <ckpt>/files_backup/*.yaml) -> load_models_dict (:65-80, `.to("cuda").eval()`) -> Shape_Prior.__init__ (:85-163: `network_dict.<part>.`
    prefix strip, load_state_dict(strict=True), FieldWrapper with sdf2occ_factor).  The checkpoint written here holds the weights the fixture was
    generated with, so the ModuleDict's prior must reproduce `shape_prior_full.npz` -- the output of the reference's OWN Shape_Prior."""
    import yaml
    from livingscenes_amd.model_utils import load_ckpt_from_log, slice_code_dict
    g = golden("shape_prior_full")
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    log = tmp_path / "log" / "shape_prior_room4cates"
    (log / "checkpoint").mkdir(parents=True)
    (log / "files_backup").mkdir()
    torch.save(synth.to_checkpoint(ew, dw, epoch=11), log / "checkpoint" / "LivingScenes_latest.pt")
    torch.save({"epoch": 3}, log / "checkpoint" / "12.pt")          # other epochs beside it are ignored by the *latest.pt glob
    field = {"model": {"model_name": "sim3sdf", "encoder_type": "vecdgcnn_atten", "decoder_type": "inner_deepsdf", "encoder": ecfg, "decoder": dcfg,
                       "sdf2occ_factor": -1.0}, "dataset": {"n_pcl": 1024}}
    (log / "files_backup" / "model_config.yaml").write_text(yaml.safe_dump(field))
    # the released configs/room4cates.yaml layout: field_pt / field_cfg placeholders (overwritten by the loader), solver_global.use_double False
    (tmp_path / "configs").mkdir()
    (tmp_path / "configs" / "room4cates.yaml").write_text(yaml.safe_dump(
        {"shape_priors": {"chair": {"field_pt": "./nowhere/selected.pt", "field_cfg": "./nowhere.yaml", "database_k": {"inv": 23}}},
         "solver_global": {"use_double": False, "use_sdf": True}}))
    monkeypatch.chdir(tmp_path)                                     # the reference reads ./configs/room4cates.yaml relative to the CWD (:268)
    models = load_ckpt_from_log(str(log))
    assert isinstance(models, torch.nn.ModuleDict) and list(models.keys()) == ["chair"]
    sp = models["chair"]
    assert not sp.training and sp.model_id == "chair" and sp.use_double is False and sp.field_input_n == 1024 and sp.cls_head is None
    assert all(p_.is_cuda for p_ in sp.parameters())
    for k, v in ew.items():                                          # prefix strip + strict load: every tensor where the reference puts it
        assert torch.equal(sp.encoder.state_dict()[k].cpu(), v), k
    x = synth.make_instances(2, 1024, seed=0).to(_dev())
    with torch.no_grad():
        emb = sp.encode(x)
        for k in ("z_so3", "z_inv", "s", "t"):
            assert emb[k].dtype == torch.float32 and tuple(emb[k].shape) == g[k].shape and relerr(emb[k], g[k]) < TOL, k
        q = torch.from_numpy(g["query"]).to(_dev())
        sdf = sp.decoder(q, None, emb, return_sdf=True)              # FieldWrapper.forward on the codes this model just produced
        assert relerr(sdf, g["sdf"]) < TOL
        occ = sp.decoder(q, None, emb)                                # default: Bernoulli(logits = sdf2occ_factor * sdf) (model_utils.py:256-263)
        assert relerr(occ.logits, -g["sdf"]) < TOL
        one = sp.decoder(q[1:], None, slice_code_dict(emb, 1), return_sdf=True)
        assert relerr(one, g["sdf"][1:]) < TOL
    # two checkpoints matching *latest.pt -> the reference's assert (:274)
    torch.save({"epoch": 0}, log / "checkpoint" / "other_latest.pt")
    with pytest.raises(AssertionError):
        load_ckpt_from_log(str(log))


def test_sdf_decode_vs_oracle_small_and_chunked():
    from oracle import net
    ecfg, dcfg = synth.small_encoder_cfg(), synth.small_decoder_cfg()
    ew, dw = synth.make_encoder_weights(ecfg, 7), synth.make_decoder_weights(dcfg, 7)
    m = _hip_model(ecfg, ew, dcfg, dw)
    g = torch.Generator().manual_seed(0)
    B, M, C = 3, 777, ecfg["c_dim"]
    code = {"z_so3": torch.randn(B, C, 3, generator=g) * 0.05, "z_inv": torch.randn(B, C, generator=g) * 0.01,
            "s": torch.rand(B, generator=g) + 0.5, "t": torch.randn(B, 1, 3, generator=g)}
    q = torch.randn(B, M, 3, generator=g)
    ref = net.field_query(dw, dcfg, q, code)
    d = _dev()
    sdf = m.sdf_decode(q.to(d), code["z_so3"].to(d), code["z_inv"].to(d), code["s"].to(d), code["t"].to(d))
    assert relerr(sdf, ref) < TOL
    sdf2 = m.sdf_decode(q.to(d), code["z_so3"].to(d), code["z_inv"].to(d), code["s"].to(d), code["t"].to(d), max_ws_bytes=1 << 18)
    assert torch.equal(sdf, sdf2)  # chunking must not change a single bit


def test_equivariance_properties():
    """The reference's own self-check (vec_dgcnn_atten.py:276-320) as assertions: z_so3 rotates with the input, z_inv is invariant, scale
    is linear in s.  The bar is 1e-4 whenever the rotated cloud produces the same graphs (k-NN of every layer, FPS of every level) -- the
    measured error is then 5e-6 .. 1.5e-5 (scripts/diag/equivariance_probe.py).  A rotation changes the rounding of the coordinates, so a
    near-tie may resolve differently and, with random weights, cascade through the layers (seed 23 of the probe: 5 % of the layer-6
    neighbours, codes 7e-2 apart -- a property of the network, the reference behaves the same): those cases keep the loose 1e-3 and only
    when at most 1 % of any layer's neighbour entries moved.  At least one of the clouds here must take the tight branch."""
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 0)
    m = _hip_model(cfg, w)
    tight = 0
    for seed in (21, 22, 25):
        x = synth.make_instances(2, 1024, seed=seed, rigid=False)
        x = x - x.mean(-1, keepdim=True)
        rng = np.random.default_rng(seed)
        R = torch.from_numpy(np.stack([synth._rand_rot(rng) for _ in range(2)]).astype(np.float32))
        sc = torch.tensor([0.7, 1.4])
        xa = torch.einsum("bij,bjn->bin", R, x * sc[:, None, None])
        z0, i0, s0, _, k0, f0 = m.encode(x.to(_dev()), pre_normalised=True, trace=True)
        z1, i1, s1, _, k1, f1 = m.encode(xa.to(_dev()), pre_normalised=True, trace=True)
        zr = torch.einsum("bij,bcj->bci", R.to(_dev()), z0)
        same = all(torch.equal(a, b) for a, b in zip(k0, k1)) and all(torch.equal(a, b) for a, b in zip(f0, f1))
        if not same:
            assert min(float((a == b).float().mean()) for a, b in zip(k0, k1)) > 0.99, seed
        tight += same
        errs = {"z_so3": relerr(z1, zr), "z_inv": relerr(i1, i0), "scale": relerr(s1, s0 * sc.to(_dev()))}
        from conftest import calibrated
        for k_, v_ in errs.items():    # same graphs: 1e-4 flat; a near-tie moved a neighbour: the measured value against its committed calibration
            calibrated(f"equivariance.seed{seed}.{'same_graphs' if same else 'moved_neighbour'}.{k_}", v_, 1e-4 if same else 1e-3)
    assert tight >= 1


# ------------------------------------------------------------------------------------------------ matcher / Kabsch / ICP
def test_greedy_match_bit_exact_on_golden_scores(golden):
    from livingscenes_amd import ops
    from oracle import more
    g = golden("matchers")
    for name in ("n1", "n2", "n3", "n5", "n32", "neg", "tie"):
        a, b = torch.from_numpy(g[f"seq_{name}_a"]), torch.from_numpy(g[f"seq_{name}_b"])
        S = more.cosine_scores(a, b)          # identical score matrix in -> identical assignment out
        m0, m1 = ops.greedy_match(S.to(_dev()))
        assert np.array_equal(m0.cpu().numpy(), g[f"seq_{name}_m0"]), name
        assert np.array_equal(m1.cpu().numpy(), g[f"seq_{name}_m1"]), name
        Sh = ops.cosine_scores(a.to(_dev()), b.to(_dev()))
        assert relerr(Sh, S) < 1e-5
    # end to end (HIP scores + HIP greedy) on the realistic 32x32 case
    a, b = torch.from_numpy(g["seq_n32_a"]).to(_dev()), torch.from_numpy(g["seq_n32_b"]).to(_dev())
    m0, m1 = ops.greedy_match(ops.cosine_scores(a, b))
    assert np.array_equal(m0.cpu().numpy(), g["seq_n32_m0"])


@pytest.mark.parametrize("n,m", [(1, 1), (1, 40), (37, 1), (32, 32), (20, 51), (51, 20), (7, 146), (40, 40), (64, 48)])
def test_greedy_match_random_shapes_bit_exact(n, m):
    """The greedy assignment loop on random rectangular score matrices -- single-wave register kernel (n * m <= 1024) and the
    workgroup kernel above that -- incl. exact ties (first row-major maximum wins), all-negative matrices (the renormalisation
    divides by a negative number: the order flips) and duplicated rows."""
    from livingscenes_amd import ops
    from oracle import more
    g = torch.Generator().manual_seed(n * 100 + m)
    for case in ("normal", "ties", "negative", "dup_rows"):
        S = torch.randn(n, m, generator=g)
        if case == "ties":
            S = torch.randint(0, 4, (n, m), generator=g).float() * 0.25
        if case == "negative":
            S = -S.abs() - 0.1
        if case == "dup_rows" and n > 1:
            S[n // 2] = S[0]
        ref = more._greedy_assign(S, n, m)
        keep = S.clone()
        m0, m1 = ops.greedy_match(S.to(_dev()))
        assert np.array_equal(m0.cpu().numpy(), ref["matches0"].numpy()), case
        assert np.array_equal(m1.cpu().numpy(), ref["matches1"].numpy()), case
        assert torch.equal(S, keep)


def test_kabsch_vs_golden(golden):
    from livingscenes_amd import ops
    g = golden("registration")
    x1, x2 = torch.from_numpy(g["kab_x1"]).to(_dev()), torch.from_numpy(g["kab_x2"]).to(_dev())
    R, t, res, fl = ops.kabsch(x1, x2, return_flags=True)
    assert relerr(R, g["kab_R"]) < TOL and relerr(t, g["kab_t"]) < TOL and relerr(res, g["kab_res"]) < TOL
    assert (fl == 0).all()
    assert (torch.det(R.cpu()) > 0.999).all()  # the reflection cases (16..31) still yield proper rotations
    Rw, tw, resw = ops.kabsch(x1, x2, torch.from_numpy(g["kab_w"]).to(_dev()))
    assert relerr(Rw, g["kab_Rw"]) < TOL and relerr(tw, g["kab_tw"]) < TOL and relerr(resw, g["kab_resw"]) < TOL
    # degenerate input (all points identical): the covariance is zero up to the eps of the weight normalisation (rank 0 or a
    # rank-1 remnant of ~1e-15).  torch.svd SUCCEEDS on such a matrix, so the reference returns a rotation with flag False; here:
    # identity (the smallest member of the least-squares family), status RANK0 / RANK1, and the high-level flag stays False
    z = torch.ones(2, 16, 3, device=_dev())
    R0, t0, _, fl0 = ops.kabsch(z, z, return_flags=True)
    assert ((fl0 == 1) | (fl0 == 2)).all() and torch.allclose(R0.cpu(), torch.eye(3).repeat(2, 1, 1), atol=1e-6)
    zero = torch.zeros(2, 16, 3, device=_dev())
    _, _, _, flz = ops.kabsch(zero, zero, return_flags=True)
    assert (flz == 2).all()
    from livingscenes_amd.lib_more.pose_estimation import kabsch_transformation_estimation
    assert not bool(kabsch_transformation_estimation(z, z)[3])
    bad = z.clone()
    bad[0, 0, 0] = float("nan")                       # non-finite input: the reference's SVD-exception branch -> identity, flag True
    Rn, tn, _, fn = kabsch_transformation_estimation(bad, z)
    assert bool(fn) and torch.equal(Rn[0].cpu(), torch.eye(3)) and float(tn[0].abs().max()) == 0.0
    # a rank-1 covariance (all points on one line): a proper rotation that maps the line direction correctly
    line = torch.linspace(-1, 1, 16, device=_dev())[None, :, None] * torch.tensor([[[1.0, 2.0, -1.0]]], device=_dev())
    Rz = torch.tensor([[[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]], device=_dev())
    moved = line @ Rz.transpose(1, 2) + 0.5
    R1, t1, res1, fl1 = ops.kabsch(line, moved, return_flags=True)
    assert int(fl1[0]) == 1 and abs(float(torch.det(R1[0].cpu())) - 1) < 1e-5 and float(res1.max()) < 1e-5
    # non-default weight paths (pose_estimation.py:58-66): best_k with weights=None, w_threshold on NORMALISED weights without renormalising
    from oracle import more
    g2 = torch.Generator().manual_seed(4)
    a, w = torch.randn(3, 40, 3, generator=g2), torch.rand(3, 40, generator=g2)
    bq = a @ torch.linalg.qr(torch.randn(3, 3, generator=g2))[0].T + 0.3
    Rk, tk, _, _ = kabsch_transformation_estimation(a.to(_dev()), bq.to(_dev()), best_k=10)       # weights=None + best_k: ones (:49-50)
    assert Rk.shape == (3, 3, 3)
    wn = w / (w.sum(1, keepdim=True) + 1e-7)
    thr = float(wn.median())
    wt = torch.where(wn < thr, torch.zeros_like(wn), wn)
    Rr, tr, _, _ = more.kabsch_transformation_estimation(a, bq, wt, normalize_w=False)
    Rh, th, _, _ = kabsch_transformation_estimation(a.to(_dev()), bq.to(_dev()), w.to(_dev()), w_threshold=thr)
    assert relerr(Rh, Rr) < TOL and relerr(th, tr) < TOL


def test_kabsch_codes_equals_kabsch_on_materialised_pseudo_points():
    """ls_kabsch_codes_f32 (z_so3 + t, the gather by matches0 and its clamp inside the launch: more_solver.py:114-116 as bench.py and
    More_Solver._register_from_codes drive it) == ls_kabsch_batched_f32 on the sums torch materialises, BIT FOR BIT; negative
    (unmatched) entries read set 0."""
    from livingscenes_amd import ops
    g = torch.Generator().manual_seed(12)
    n, c = 40, 256
    z1, z2 = torch.randn(n, c, 3, generator=g) * 0.03, torch.randn(n, c, 3, generator=g) * 0.03
    t1, t2 = torch.randn(n, 1, 3, generator=g), torch.randn(n, 1, 3, generator=g)
    d = _dev()
    R0, tt0, res0 = ops.kabsch((z1 + t1).to(d), (z2 + t2).to(d))
    R1, tt1, res1 = ops.kabsch_codes(z1.to(d), t1.to(d), z2.to(d), t2.to(d), want_res=True)
    assert torch.equal(R0, R1) and torch.equal(tt0, tt1) and torch.equal(res0, res1)
    m0 = torch.randperm(n, generator=g)
    m0[[3, 17]] = -1
    R2, tt2 = ops.kabsch_codes(z1.to(d), t1.to(d), z2.to(d), t2.to(d), sel2=m0.to(d))
    j = m0.clamp(min=0)
    R3, tt3, _ = ops.kabsch((z1 + t1).to(d), (z2 + t2).index_select(0, j).to(d))
    assert torch.equal(R2, R3) and torch.equal(tt2, tt3)
    sel1 = torch.tensor([5, 5, 0, 39])
    R4, tt4 = ops.kabsch_codes(z1.to(d), t1.to(d), z2.to(d), t2.to(d), sel1=sel1.to(d), sel2=torch.tensor([1, 2, 3, 4]).to(d))
    R5, tt5, _ = ops.kabsch((z1 + t1)[sel1].to(d), (z2 + t2)[[1, 2, 3, 4]].to(d))
    assert torch.equal(R4, R5) and torch.equal(tt4, tt5)


def test_residual_matrix_and_eq_matchers(golden):
    from livingscenes_amd import ops
    from oracle import more
    g = golden("matchers")
    src, tgt = torch.from_numpy(g["eqsrc_z_so3"]), torch.from_numpy(g["eqtgt_z_so3"])
    ref = more.kabsch_residual_matrix(src, tgt)
    res = ops.kabsch_residual_matrix(src.to(_dev()), tgt.to(_dev()))
    assert relerr(res, ref) < TOL
    m0, _ = ops.greedy_match((1 / (res + 1e-5)))
    assert np.array_equal(m0.cpu().numpy(), g["eq_m0"])


def test_icp_vs_oracle():
    """ls_icp_f32 vs the oracle's restatement of pytorch3d.ops.iterative_closest_point (PARITY UNPINNED: pytorch3d is absent),
    problem by problem (the kernel stops every problem on its own, i.e. it equals independent batch-1 calls).  ICP is a
    fixed-point iteration on a DISCRETE nearest-neighbour assignment: the two sides agree to 1e-4 exactly when they end on the
    same assignment, and a single neighbour that flips in the last iterations (distances equal to ~1e-7) moves the least-squares
    pose by ~1/n.  So: iteration counts within 1, final assignments >= 99.5 % equal, pose within 1e-4 where the assignment is
    identical (required for most problems) and within 1e-3 otherwise."""
    from livingscenes_amd import ops
    from oracle import more
    P = 6
    sc = synth.make_scene_pair(P, 1024, seed=3, noise=0.002)
    X, Y = sc["ref"], sc["rescan"]
    gt = more.se3_concatenate(sc["rescan_T"][:, :3], more.se3_inverse(sc["ref_T"][:, :3]))
    rng = np.random.default_rng(1)
    # perturbed ground truth as initial guess (column convention) -> row convention for ICP
    dR = torch.from_numpy(np.stack([synth._rand_rot(rng) for _ in range(P)]).astype(np.float32))
    dR = torch.matrix_exp(0.05 * (dR - dR.transpose(1, 2)))
    R0 = (dR @ gt[:, :, :3]).transpose(1, 2).contiguous()
    T0 = (gt[:, :, 3] + 0.02).contiguous()
    R, T, rmse, iters = ops.icp(X.to(_dev()), Y.to(_dev()), R0.to(_dev()), T0.to(_dev()))
    R, T, iters = R.cpu(), T.cpu(), iters.cpu()
    exact = 0
    for p in range(P):
        Rp, Tp, rp, ip, _ = more.iterative_closest_point(X[p:p + 1], Y[p:p + 1], R0[p:p + 1], T0[p:p + 1])
        assert abs(int(iters[p]) - ip) <= 1, (p, int(iters[p]), ip)
        nn_h = more._nn1(X[p:p + 1] @ R[p:p + 1] + T[p:p + 1, None], Y[p:p + 1])
        nn_o = more._nn1(X[p:p + 1] @ Rp + Tp[:, None], Y[p:p + 1])
        agree = float((nn_h == nn_o).float().mean())
        assert agree >= 0.995, (p, agree)
        exact += agree == 1.0
        from conftest import calibrated
        kind = "same_assignment" if agree == 1.0 else "flipped_neighbour"
        calibrated(f"icp.problem{p}.{kind}.R", relerr(R[p:p + 1], Rp), 1e-4 if agree == 1.0 else 1e-3)
        calibrated(f"icp.problem{p}.{kind}.T", relerr(T[p:p + 1], Tp), 1e-4 if agree == 1.0 else 1e-3)
        assert abs(float(rmse[p]) - float(rp)) < 1e-4 * max(float(rp), 1e-3)
    assert exact >= P - 1, f"only {exact} of {P} problems ended on the oracle's nearest-neighbour assignment"


def test_solve_R_on_the_device_vs_the_reference_formula():
    """pose_estimation.solve_R (pose_estimation.py:11-27; used by solve_transform_from_latent) no longer goes through torch.svd on the host: the
    mirrored point sets through the device Kabsch kernel against the reference's formula evaluated in fp64 -- rotations of noisy z_so3-like
    sets, a reflection case (det < 0 before the fix-up) and the un-batched [m, 3] call."""
    from livingscenes_amd.lib_more.pose_estimation import solve_R, solve_transform_from_latent
    g = torch.Generator().manual_seed(3)
    b, m = 6, 256
    f1 = torch.randn(b, m, 3, generator=g)
    rng = np.random.default_rng(5)
    Rt = torch.from_numpy(np.stack([synth._rand_rot(rng) for _ in range(b)]).astype(np.float32))
    f2 = f1 @ Rt.transpose(1, 2) + 0.01 * torch.randn(b, m, 3, generator=g)
    f2[1] = f2[1] * torch.tensor([1.0, 1.0, -1.0])        # a mirrored target: the SVD solution has det < 0 before the diag(1, 1, d) fix-up

    def ref(a, c):
        H = a.double().transpose(-1, -2) @ c.double()
        U, _, Vh = torch.linalg.svd(H)
        V = Vh.transpose(-1, -2)
        d = torch.linalg.det(V @ U.transpose(-1, -2))
        D = torch.diag_embed(torch.stack([torch.ones_like(d), torch.ones_like(d), d], -1))
        return V @ D @ U.transpose(-1, -2)
    R = solve_R(f1.to(_dev()), f2.to(_dev()))
    assert R.shape == (b, 3, 3) and R.is_cuda
    assert relerr(R, ref(f1, f2)) < TOL
    assert (torch.linalg.det(R.cpu().double()) - 1).abs().max() < 1e-5
    assert relerr(solve_R(f1[0].to(_dev()), f2[0].to(_dev())), ref(f1[0], f2[0])) < TOL
    c1 = {"z_so3": f1[:1].to(_dev()), "t": torch.randn(1, 1, 3, generator=g).to(_dev())}
    c2 = {"z_so3": f2[:1].to(_dev()), "t": torch.randn(1, 1, 3, generator=g).to(_dev())}
    T = solve_transform_from_latent(c1, c2)
    assert T.shape == (1, 4, 4) and relerr(T[:, :3, :3], ref(f1[:1], f2[:1])) < TOL


def test_deepsdf_decoder_direct_forward_vs_oracle():
    """DeepSDF_Decoder.forward(input [B,M,513], 'val') -- the reference class's own call surface (deepsdf_decoder.py:78-123), not
    only the fused FieldWrapper route -- against oracle.net.decoder_forward, released widths and the small config."""
    from livingscenes_amd.deepsdf_decoder import DeepSDF_Decoder
    from oracle import net
    for dcfg in (synth.small_decoder_cfg(), synth.default_decoder_cfg()):
        dw = synth.make_decoder_weights(dcfg, 2)
        dec = DeepSDF_Decoder(**dcfg)
        dec.load_state_dict(dw, strict=True)
        dec = dec.to(_dev())
        g = torch.Generator().manual_seed(9)
        inp = torch.randn(2, 257, dcfg["latent_size"] + dcfg["pe_dim"], generator=g) * 0.2
        ref = net.decoder_forward(dw, dcfg, inp)
        with torch.no_grad():
            out = dec(inp.to(_dev()), "val")
        assert out.shape == (2, 257) and relerr(out, ref) < TOL
    with pytest.raises(NotImplementedError):
        dec(inp.to(_dev()), "train")


def test_sdf_dense_grid_mise_resolution():
    """A 33^3 grid (MISE resolution0 + 1, mesh_extractor2.py:100-105) for 2 instances through the chunked decoder path."""
    from oracle import net
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    m = _hip_model(ecfg, ew, dcfg, dw)
    g = torch.Generator().manual_seed(5)
    B, C = 2, 256
    code = {"z_so3": torch.randn(B, C, 3, generator=g) * 0.03, "z_inv": torch.randn(B, C, generator=g) * 0.003,
            "s": torch.ones(B), "t": torch.zeros(B, 1, 3)}
    lin = torch.linspace(-0.55, 0.55, 33)
    q = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(1, -1, 3).expand(B, -1, -1).contiguous()
    ref = net.field_query(dw, dcfg, q, code)
    d = _dev()
    sdf = m.sdf_decode(q.to(d), code["z_so3"].to(d), code["z_inv"].to(d), code["s"].to(d), code["t"].to(d), max_ws_bytes=64 << 20)
    assert sdf.shape == (B, 33 ** 3) and relerr(sdf, ref) < TOL


# ------------------------------------------------------------------------------------------------ decoder backward
@pytest.mark.parametrize("which,B,M", [("small", 2, 300), ("small", 1, 1024), ("full", 2, 512)])
def test_sdf_backward_vs_oracle_autograd(which, B, M):
    """ls_sdf_decode_train / ls_sdf_backward vs torch autograd through the oracle's FieldWrapper + DeepSDF restatement (CPU):
    gradients of a random linear functional of the SDF w.r.t. query, z_so3, z_inv, s, t within TOL of their max-norm."""
    from livingscenes_amd import ops, packing
    from oracle import net
    ecfg, dcfg = (synth.small_encoder_cfg(), synth.small_decoder_cfg()) if which == "small" else (synth.default_encoder_cfg(), synth.default_decoder_cfg())
    ew, dw = synth.make_encoder_weights(ecfg, 3), synth.make_decoder_weights(dcfg, 3)
    desc, blob = packing.pack_model(ew, ecfg, dw, dcfg)
    m = ops.HipModel(desc, blob, _dev())
    g = torch.Generator().manual_seed(B * 31 + M)
    L = dcfg["latent_size"]
    code = {"z_so3": torch.randn(B, L, 3, generator=g) * 0.05, "z_inv": torch.randn(B, L, generator=g) * 0.05,
            "s": torch.rand(B, generator=g) * 0.5 + 0.75, "t": torch.randn(B, 1, 3, generator=g) * 0.1}
    q = synth.make_queries(B, M, seed=5) * code["s"][:, None, None] + code["t"]
    q[0, 0] = code["t"][0, 0]                      # a query exactly at the instance origin: |q| = 0, norm gradient defined as 0
    gsdf = torch.randn(B, M, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in code.items()}
    ql = q.clone().requires_grad_(True)
    sdf_ref = net.field_query_with_grad(dw, dcfg, ql, leaves)
    (sdf_ref * gsdf).sum().backward()
    dev = _dev()
    sdf, saved = m.sdf_decode_train(q.to(dev), code["z_so3"].to(dev), code["z_inv"].to(dev), code["s"].to(dev), code["t"].to(dev))
    assert relerr(sdf, sdf_ref.detach()) < TOL
    gq, gso3, ginv, gs, gt = m.sdf_backward(saved, gsdf.to(dev))
    assert relerr(gso3, leaves["z_so3"].grad) < TOL
    assert relerr(ginv, leaves["z_inv"].grad) < TOL
    assert relerr(gt, leaves["t"].grad.reshape(B, 3)) < TOL
    assert relerr(gs, leaves["s"].grad) < TOL
    assert relerr(gq, ql.grad) < TOL


# ------------------------------------------------------------------------------------------------ Sinkhorn (UNPINNED)
@pytest.mark.parametrize("N,M,shift", [(1024, 1024, 0.05), (300, 517, 0.3), (64, 64, 0.0)])
def test_sinkhorn_divergence_vs_oracle(N, M, shift):
    """csrc/sinkhorn.hip + the epsilon-scaling loop vs the torch-CPU restatement of the same definition (PARITY UNPINNED w.r.t.
    geomloss, see oracle/sinkhorn.py): loss and its gradient w.r.t. the source points; identical clouds give ~0."""
    from livingscenes_amd.sinkhorn import sinkhorn_divergence
    from oracle import sinkhorn as osk
    g = torch.Generator().manual_seed(N + M)
    y = torch.rand(M, 3, generator=g) - 0.5
    x = (y[torch.randperm(M, generator=g)[:N] % M] if N <= M else torch.rand(N, 3, generator=g) - 0.5) + shift * torch.randn(N, 3, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = osk.sinkhorn_divergence(xr, y)
    ref.backward()
    xh = x.to(_dev()).requires_grad_(True)
    got = sinkhorn_divergence(xh[None], y.to(_dev())[None])
    got.backward()
    scale = max(abs(float(ref)), 1e-6)
    assert abs(float(got) - float(ref)) < 2e-4 * scale + 2e-7
    assert relerr(xh.grad, xr.grad) < 5e-4 or float(xr.grad.abs().max()) < 1e-7
    if shift == 0.0:
        assert abs(float(got)) < 1e-5
