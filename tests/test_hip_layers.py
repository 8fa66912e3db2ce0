"""Isolated GPU parity tests of the encoder's layer operators exported by the C ABI (SURVEY.md 8 b: ls_vn_edgeconv_pool_f32,
ls_vn_edgeconv_attn_f32, ls_vn_lna_f32, ls_encoder_tail_f32) -- rows a-7 (VecActivation) and a-8 (VecLNA / cevn / VecResBlock)
of the scope table, which the end-to-end encoder tests only cover in composition.

Two references:
  * tests/golden/encoder_small_layers.npz -- per-layer tensors recorded from the reference's OWN VecDGCNN_att.forward by hooks
    (tests/golden/make_golden.py section 2b): layer inputs, k-NN / FPS indices, attention messages, global-conv outputs, heads;
  * oracle.net (pinned against tests/golden/vn_layers.npz on the CPU) at the released widths.
Reference: lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:186-250, vec_layers.py:24-31,121-136,241-268,523-534,631-651.
"""
import numpy as np
import pytest
import torch

from livingscenes_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4  # of the reference tensor's max-norm


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


def _hip(cfg, w):
    from livingscenes_amd import ops, packing
    desc, blob = packing.pack_model(w, cfg, None, None)
    return ops.HipModel(desc, blob, _dev())


def test_layer_operators_vs_reference_layer_fixture(golden):
    """Every layer operator on the tensors the REFERENCE itself produced at that point of its forward pass."""
    g = golden("encoder_small_layers")
    cfg = synth.small_encoder_cfg()
    m = _hip(cfg, synth.make_encoder_weights(cfg, 7))
    d = _dev()
    L, g0, ds = cfg["num_layers"], cfg["res_global_start_layer"], cfg["down_sample_layers"]
    T = lambda k: torch.from_numpy(g[k]).to(d)
    level = 0
    for i in range(L):
        rows_i = None
        if i in ds:
            rows_i = T(f"fps_idx_{level}")
            level += 1
        src = T(f"src_{i}").reshape(2, -1, 3) if i == 0 else T(f"src_{i}")
        msg = m.edgeconv(i, src, T(f"knn_idx_{i}"), rows_i)
        want_msg = g[f"msg_{i}"] if i >= g0 else g[f"src_{i + 1}"]
        assert relerr(msg, want_msg) < TOL, f"edge-conv layer {i}"
        if i >= g0:
            out = m.vn_lna_global(i, T(f"msg_{i}"))
            assert relerr(out, g[f"out_{i}"]) < TOL, f"global conv layer {i}"
    z_so3, z_inv, s, t = m.encoder_tail(T(f"out_{L - 1}"))
    assert relerr(z_so3, g["z_so3"]) < TOL and relerr(z_inv, g["z_inv"]) < TOL
    assert relerr(s, g["scale"]) < TOL and relerr(t, g["center"].reshape(2, 3)) < TOL


@pytest.mark.parametrize("B,N", [(2, 1024), (1, 1000), (3, 936)])
def test_layer_operators_vs_oracle_released_widths(B, N):
    """The released configuration (widths 32 .. 512, three down-sampling layers) layer by layer against oracle.net's trace; the
    ragged sizes leave partial workgroups / tiles in the fused kernels (16 or 8 points per workgroup, 120-row GEMM tiles)."""
    from oracle import net
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 0)
    x = synth.make_instances(B, N, seed=11, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.2
    tr = {}
    center, scale, z_so3, z_inv = net.encoder_forward(w, cfg, x, trace=tr)
    m = _hip(cfg, w)
    d = _dev()
    L, g0, ds = cfg["num_layers"], cfg["res_global_start_layer"], cfg["down_sample_layers"]
    for i in range(L):
        src = rows(tr[f"src_f_{i}"]).to(d)
        if i == 0:
            src = src.reshape(B, N, 3)
        rows_i = tr[f"fps_idx_{i}"].to(torch.int32).to(d) if i in ds else None
        msg = m.edgeconv(i, src, tr[f"knn_idx_{i}"].to(torch.int32).to(d), rows_i)
        assert relerr(msg, rows(tr[f"msg_f_{i}"])) < TOL, f"edge-conv layer {i}"
        if i >= g0:
            out = m.vn_lna_global(i, rows(tr[f"msg_f_{i}"]).to(d))
            assert relerr(out, rows(tr[f"dst_f_{i}"])) < TOL, f"global conv layer {i}"
    hz, hi, hs, ht = m.encoder_tail(rows(tr[f"dst_f_{L - 1}"]).to(d))
    assert relerr(hz, z_so3) < TOL and relerr(hi, z_inv) < TOL and relerr(hs, scale) < TOL and relerr(ht, center.reshape(B, 3)) < TOL
    # encode epilogue (model_utils.py:182-185): t = center + centroid, s = scale_0 * scale
    cen, sc0 = torch.randn(B, 3), torch.rand(B) + 0.5
    ez, ei, es, et = m.encoder_tail(rows(tr[f"dst_f_{L - 1}"]).to(d), cen.to(d), sc0.to(d))
    assert torch.equal(ez, hz) and torch.equal(ei, hi)
    assert relerr(es, sc0 * scale) < TOL and relerr(et, center.reshape(B, 3) + cen) < TOL


def test_vn_activation_and_cevn_closed_forms_vs_oracle():
    """a-7 / a-8 directly: the global conv operator IS VecLNA(cat(f, mean f)) and the tail IS cevn + VecResBlock; feed inputs
    that exercise the activation's both branches (negative and positive <x, k>), zero vectors and badly scaled channels."""
    from oracle import net
    cfg = synth.small_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 5)
    m = _hip(cfg, w)
    d = _dev()
    i = cfg["res_global_start_layer"]
    C = cfg["feat_dim"][i]
    gen = torch.Generator().manual_seed(3)
    f = torch.randn(3, C, 3, 70, generator=gen)
    f[0, :, :, 5] = 0.0                               # a zero 3-vector in every channel: the 1e-12 clamps decide
    f[1, :4] *= 1e4                                   # channels of very different magnitude
    f[2, :, :, :10] *= 1e-6
    j = i - cfg["res_global_start_layer"]
    gmean = f.mean(-1, keepdim=True).expand_as(f)
    ref = net.vec_lna(torch.cat([f, gmean], 1), w[f"global_conv_list.{j}.lin.weight"], w[f"global_conv_list.{j}.act.lin_dir.weight"], 0.2)
    out = m.vn_lna_global(i, rows(f).to(d))
    assert relerr(out, rows(ref)) < TOL
    # per-instance accuracy too (the badly scaled instances must not hide behind the large one)
    for b in range(3):
        assert relerr(out[b], rows(ref)[b]) < TOL, b


def test_attention_layer_is_reproducible_with_streams_in_flight():
    """The attention edge-conv on identical inputs from 8 streams at once (other kernels of the same launch sequence in flight on
    the same CUs): every output must be bit-identical to the serial result."""
    from oracle import net
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 0)
    B, N = 16, 1024
    x = synth.make_instances(B, N, seed=21, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.2
    m = _hip(cfg, w)
    d = _dev()
    z = m.encode(x.to(d), pre_normalised=True, trace=True)
    knn_l, fps_l = z[4], z[5]
    # layer-2 inputs from the HIP path itself: features after layer 1
    f1 = m.edgeconv(1, m.edgeconv(0, x.transpose(1, 2).contiguous().to(d), knn_l[0]), knn_l[1])
    ref = m.edgeconv(2, f1, knn_l[2], fps_l[0])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=d) for _ in range(8)]
    for rep in range(3):
        outs = []
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(d))
            with torch.cuda.stream(s):
                outs.append(m.edgeconv(2, f1, knn_l[2], fps_l[0]))
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert torch.equal(o, ref), f"rep {rep} stream {k}: {int((o != ref).sum())} floats differ"


def test_table_free_32_point_layers_match_the_table_path():
    """Layers 5 / 6 of the released schedule (32 destination points; 128 / 32 source points) WITHOUT a table (csrc/edge_fused.hip: the table
    slices formed in LDS by two launches that exchange per-head partial norms) against the table GEMM + edge_attn_v4_kernel pair
    (ls_model_set_option(LS_OPT_EDGE_FUSE_T, 0)).  Same products (two-piece f16 split, ascending k, same term order); what differs is the order in
    which the squared norms are summed over the channels and the exact row maximum behind each row's power-of-two scale, so the outputs agree to
    fp32 round-off, not bit for bit: 2e-6 of the tensor maximum.  Batches of 1, 3 and 64 instances (one workgroup per (instance, head group)).
    The row maxima the table-free kernels hand to the global conv's GEMM (one per head group: GemmAux) are exercised by running layer + global
    conv through ls_encode in both modes (second half)."""
    from livingscenes_amd import _lib
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 0)
    m = _hip(cfg, w)
    d = _dev()
    g = torch.Generator().manual_seed(9)
    for B in (1, 3, 64):
        for layer, Ns, Nd, Cin in ((5, 128, 32, 128), (6, 32, 32, 256)):
            src = torch.randn(B, Ns, 3, Cin, generator=g).to(d)
            knn = torch.stack([torch.stack([torch.randperm(Ns, generator=g)[:16] for _ in range(Nd)]) for _ in range(B)]).to(torch.int32).to(d)
            rows_i = torch.stack([torch.randperm(Ns, generator=g)[:Nd] for _ in range(B)]).to(torch.int32).to(d) if Ns != Nd else None
            a = m.edgeconv(layer, src, knn, rows_i).cpu().numpy()
            prev = m.set_option(_lib.OPT_EDGE_FUSE_T, 0)
            b = m.edgeconv(layer, src, knn, rows_i).cpu().numpy()
            m.set_option(_lib.OPT_EDGE_FUSE_T, prev)
            assert np.isfinite(a).all() and a.shape == b.shape
            assert np.abs(a - b).max() / np.abs(b).max() <= 2e-6, (layer, B)
    # whole encoder, both modes: the per-head-group row maxima feed the scaling of the global conv's GEMM operands
    x = synth.make_instances(3, 1024, seed=21).to(d)
    codes = {}
    for mode in (1, 0):
        prev = m.set_option(_lib.OPT_EDGE_FUSE_T, mode)
        codes[mode] = [t.clone() for t in m.encode(x)]
        m.set_option(_lib.OPT_EDGE_FUSE_T, prev)
    for a, b in zip(codes[1], codes[0]):
        assert relerr(a, b) < 2e-5, relerr(a, b)


@pytest.mark.parametrize("B", [2, 3])
def test_staged_attention_layers_vs_oracle_and_the_gather_kernel(B):
    """Attention layers 2 - 4 at the released widths through the LDS-staged kernel (edge_staged.hip: slice-major table, slices streamed through the
    LDS, ls_model_set_option(LS_OPT_EDGE_STAGED, 2) = whenever the shape fits) against oracle.net (1e-4 of the tensor maximum) and against the
    row-gather kernel (edge_attn_fq_kernel, option 0).  Same products and the same arithmetic per edge and channel; the head / norm sums run in a
    different order (per-channel chains + lane trees instead of four-channel chains), so the two kernels agree to 2e-6, not bit for bit
    (vec_dgcnn_atten.py:205-219)."""
    from oracle import net
    from livingscenes_amd import _lib
    N = 1024
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 3)
    x = synth.make_instances(B, N, seed=5, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.2
    tr = {}
    net.encoder_forward(w, cfg, x, trace=tr)
    m = _hip(cfg, w)
    d = _dev()
    ds = cfg["down_sample_layers"]
    prev = m.get_option(_lib.OPT_EDGE_STAGED)
    for i in (2, 3, 4):
        src = rows(tr[f"src_f_{i}"]).to(d)
        rows_i = tr[f"fps_idx_{i}"].to(torch.int32).to(d) if i in ds else None
        knn = tr[f"knn_idx_{i}"].to(torch.int32).to(d)
        m.set_option(_lib.OPT_EDGE_STAGED, 2)
        staged = m.edgeconv(i, src, knn, rows_i).clone()
        again = m.edgeconv(i, src, knn, rows_i).clone()
        m.set_option(_lib.OPT_EDGE_STAGED, 0)
        gathered = m.edgeconv(i, src, knn, rows_i).clone()
        m.set_option(_lib.OPT_EDGE_STAGED, prev)
        assert torch.equal(staged, again), f"layer {i}: the staged kernel is not reproducible"
        assert relerr(staged, rows(tr[f"msg_f_{i}"])) < TOL, f"staged attention layer {i} vs oracle"
        assert relerr(staged, gathered) < 2e-6, f"staged vs gather kernel, layer {i}: {relerr(staged, gathered)}"


def test_staged_attention_inside_the_encoder_vs_oracle():
    """ls_encode with the staged attention forced on (LS_OPT_EDGE_STAGED = 2; a 2-instance batch is far below the automatic threshold) against
    oracle.net run on the DEVICE's k-NN graph (features that differ by round-off may flip a feature-space near-tie; the flips themselves are
    audited by test_hip_parity.py::test_encoder_forward_vs_oracle): FPS identical, codes within 1e-4 of their maxima."""
    from oracle import net
    from livingscenes_amd import _lib
    cfg = synth.default_encoder_cfg()
    w = synth.make_encoder_weights(cfg, 1)
    m = _hip(cfg, w)
    B, N = 2, 1024
    x = synth.make_instances(B, N, seed=9, rigid=False)
    x = (x - x.mean(-1, keepdim=True)) / 1.1
    prev = m.set_option(_lib.OPT_EDGE_STAGED, 2)
    hz, hi, hs, ht, knn_l, fps_l = m.encode(x.to(_dev()), pre_normalised=True, trace=True)
    m.set_option(_lib.OPT_EDGE_STAGED, prev)
    tr = {}
    graph = {i: knn_l[i].cpu() for i in range(1, cfg["num_layers"])}
    center, scale, z_so3, z_inv = net.encoder_forward(w, cfg, x, trace=tr, graph=graph)
    for j, i in enumerate(cfg["down_sample_layers"]):
        assert np.array_equal(fps_l[j].cpu().numpy(), tr[f"fps_idx_{i}"].numpy().astype(np.int32)), f"fps level {j}"
    assert relerr(hz, z_so3) < TOL and relerr(hi, z_inv) < TOL
    assert relerr(hs, scale) < TOL and relerr(ht, center.squeeze(1)) < TOL
