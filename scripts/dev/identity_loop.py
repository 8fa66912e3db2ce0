"""dev: bit-identity of 12 model handles in flight on the bench batch, R rounds (the reproducibility defect of DESIGN 4.3 showed as 1e-6 differences between
handles).    python scripts/dev/identity_loop.py [rounds]      (LS_LIB_PATH selects the library)   -> "rounds R handle-results H mismatches M" """
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from livingscenes_amd import _lib, synth  # noqa: E402
from livingscenes_amd.model_utils import Shape_Prior  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
ew, dw = synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0)
sps = [Shape_Prior.from_state(ecfg, dcfg, ew, dw, device=dev) for _ in range(12)]
for s in sps:
    s.hip_model().set_option(_lib.OPT_GEMM_OVERLAP, 0)
streams = [torch.cuda.Stream(device=dev) for _ in range(12)]
scene = synth.make_scene_pair(32, 1024, seed=1000)
x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)
torch.cuda.synchronize()
ref, bad, total = None, 0, 0
per_key = {k: 0 for k in ("z_so3", "z_inv", "s", "t")}
worst = {k: 0.0 for k in per_key}
with torch.no_grad():
    for r in range(rounds):
        outs = []
        for i in range(12):
            with torch.cuda.stream(streams[i]):
                outs.append(sps[i].encode(x))
        torch.cuda.synchronize()
        if ref is None:
            ref = {k: v.clone() for k, v in outs[0].items()}
        for o in outs:
            total += 1
            bad += int(not all(torch.equal(o[k], ref[k]) for k in ("z_so3", "z_inv", "s", "t")))
            for k in per_key:
                if not torch.equal(o[k], ref[k]):
                    per_key[k] += 1
                    worst[k] = max(worst[k], float((o[k] - ref[k]).abs().max() / ref[k].abs().max()))
print(f"rounds {rounds} handle-results {total} mismatches {bad}" + ("" if not bad else f" per output {per_key} worst relative difference {worst}"))
