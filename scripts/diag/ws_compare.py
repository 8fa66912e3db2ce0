"""Race hunting: run the first n encoder layers (LS_DEBUG_LAYERS=n) serially and with 8 handles in flight, and compare the
WORKSPACES region by region (regions from the LS_PLAN line the library prints)."""
import os, sys, re, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    os.environ.setdefault("LS_GEMM_OVERLAP", "0")
    import torch
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
    from livingscenes_amd import synth
    from livingscenes_amd.model_utils import Shape_Prior
    dev = torch.device("cuda:0")
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    nfl = 8
    W = lambda h: max(h._ws.values(), key=lambda t: t.numel())
    sps = [Shape_Prior.from_state(ecfg, dcfg, ew, dw, device=dev) for _ in range(nfl)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    scene = synth.make_scene_pair(32, 1024, seed=1000)
    x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)
    with torch.no_grad():
        hs = [sp.hip_model() for sp in sps]
        for h in hs: h.encode(x)
        torch.cuda.synchronize()
        plan = dict(knn=1737728, knn2=5932032, knns=10126336, hint=115796992, inv=119991296, fA=120253440, fB=145419264, msg=170585088,
                    T=195750912, TG=548072448, g=598404096, G=598797312, Tc=600370176, gws=606759936, total=631926016)
        names = list(plan)
        W(hs[0]).zero_(); hs[0].encode(x); torch.cuda.synchronize()
        ref = W(hs[0]).clone()
        def report(tag, ws):
            out = []
            for a, b in zip(names[:-1], names[1:]):
                if a == "knns":
                    continue
                d = int((ws[plan[a]:plan[b]] != ref[plan[a]:plan[b]]).sum())
                if d:
                    out.append(f"{a}:{d}")
                    if a in ("msg", "T", "knn", "knn2") and os.environ.get("LS_DIAG_VERBOSE"):
                        fa = ws[plan[a]:plan[b]].view(torch.float32); fr = ref[plan[a]:plan[b]].view(torch.float32)
                        idx = (fa != fr).nonzero().flatten()
                        print(f"   {a}: {idx.numel()} floats differ; first idx {idx[:8].tolist()} last {int(idx[-1])}; got {fa[idx[:4]].tolist()} ref {fr[idx[:4]].tolist()}; points {sorted(set((idx // 192).tolist()))[:10]}")
            print(tag, " ".join(out) if out else "identical")
        W(hs[0]).zero_(); hs[0].encode(x); torch.cuda.synchronize()
        report("serial repeat:", W(hs[0]))
        for rep in range(2):
            for h in hs: W(h).zero_()
            torch.cuda.synchronize()
            for i in range(nfl):
                with torch.cuda.stream(streams[i]):
                    hs[i].encode(x)
            torch.cuda.synchronize()
            for i in range(nfl):
                report(f"rep {rep} handle {i}:", W(hs[i]))
    sys.exit(0)
for n in sys.argv[1:]:
    print("=== LS_DEBUG_LAYERS =", n, flush=True)
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LS_DEBUG_LAYERS=n), capture_output=True, text=True)
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l)[-3000:], flush=True)
