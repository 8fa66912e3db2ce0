"""Worker of tests/test_hip_fullbatch.py: runs the EXACT bench.py batch (BASELINE configs[1]+[2]: a 32-object scene and its
rescan = 64 instances x 1024 points, seed 1000) the way bench.py runs it -- GPU_MAX_HW_QUEUES=16 (must be in the environment
before the HIP runtime starts, hence a separate process), N model handles on N streams, all in flight at once -- and writes
the codes of every handle, the per-layer k-NN / FPS index lists of one traced pass, the matcher output and the Kabsch poses to
an .npz for the parent process to compare with the oracle.

    python tests/tools/fullbatch_worker.py OUT.npz [n_obj] [N] [handles]
"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402
import torch  # noqa: E402

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def main():
    out_path = sys.argv[1]
    n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    nfl = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    from livingscenes_amd import synth
    from livingscenes_amd.lib_more.matcher_new import sequential_matcher
    from livingscenes_amd.lib_more.pose_estimation import kabsch_transformation_estimation
    from livingscenes_amd.model_utils import Shape_Prior
    dev = torch.device("cuda:0")
    ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
    ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
    sp = Shape_Prior.from_state(ecfg, dcfg, ew, dw, device=dev)
    sps = [sp] + [Shape_Prior.from_state(ecfg, dcfg, sp.encoder.state_dict(), sp.decoder.F.state_dict(), device=dev)
                  for _ in range(nfl - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    from livingscenes_amd import _lib
    for s_ in sps:                                             # as bench.py with steps in flight
        s_.hip_model().set_option(_lib.OPT_GEMM_OVERLAP, 0)
    scene = synth.make_scene_pair(n_obj, N, seed=1000)
    x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)

    def step(sp_):
        emb = sp_.encode(x)
        m = sequential_matcher(emb["z_inv"][:n_obj], emb["z_inv"][n_obj:])
        j = m["matches0"].clamp(min=0)
        p1 = emb["z_so3"][:n_obj] + emb["t"][:n_obj]
        p2 = (emb["z_so3"][n_obj:] + emb["t"][n_obj:]).index_select(0, j)
        R, t, _, _ = kabsch_transformation_estimation(p1, p2)
        return emb, m, R, t

    outs = [None] * nfl
    with torch.no_grad():
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        for rep in range(3):                       # three rounds: workspaces are reused while other handles are in flight
            for i in range(nfl):
                with torch.cuda.stream(streams[i]):
                    outs[i] = step(sps[i])
        torch.cuda.synchronize()
        z_so3, z_inv, s, t, knn_l, fps_l = sp.hip_model().encode(x, trace=True)
        torch.cuda.synchronize()
    res = {"x": x.cpu().numpy(), "hw_queues": np.int32(os.environ["GPU_MAX_HW_QUEUES"]), "handles": np.int32(nfl)}
    for i, (emb, m, R, tt) in enumerate(outs):
        for k, v in emb.items():
            res[f"h{i}_{k}"] = v.cpu().numpy()
        res[f"h{i}_m0"], res[f"h{i}_m1"] = m["matches0"].cpu().numpy(), m["matches1"].cpu().numpy()
        res[f"h{i}_pose_R"], res[f"h{i}_pose_t"] = R.cpu().numpy(), tt.cpu().numpy()
    res["tr_z_so3"], res["tr_z_inv"], res["tr_s"], res["tr_t"] = (v.cpu().numpy() for v in (z_so3, z_inv, s, t))
    for i, k in enumerate(knn_l):
        res[f"knn_{i}"] = k.cpu().numpy()
    for i, f in enumerate(fps_l):
        res[f"fps_{i}"] = f.cpu().numpy()
    np.savez(out_path, **res)
    print("fullbatch worker ok")


if __name__ == "__main__":
    main()
