"""diag (round 6): which pointwise.hip kernel is not reproducible when the file is built WITH the SLP vectoriser (scripts/dev/pk_guard_ab.sh: the whole
encode differs by percents from call to call)?  Each operator alone, R repetitions on one stream and on 8 streams, compared with its first result.
    LS_LIB_PATH=.../variants/pk_pw_slp/liblivingscenes_hip.so python scripts/diag/pointwise_pk_isolate.py [R]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from livingscenes_amd import synth  # noqa: E402
from livingscenes_amd.model_utils import Shape_Prior  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), synth.make_decoder_weights(dcfg, 0), device=dev)
m = sp.hip_model()
from livingscenes_amd import ops  # noqa: E402

scene = synth.make_scene_pair(32, 1024, seed=1000)
x = torch.cat([scene["ref"], scene["rescan"]], 0).transpose(1, 2).contiguous().to(dev)
g = torch.Generator().manual_seed(3)
f_tail = (torch.randn(64, 32, 3, 512, generator=g) * 0.05).to(dev)
f_glob = {2: (torch.randn(64, 512, 3, 64, generator=g) * 0.1).to(dev), 4: (torch.randn(64, 128, 3, 128, generator=g) * 0.1).to(dev),
          6: (torch.randn(64, 32, 3, 512, generator=g) * 0.1).to(dev)}
cases = {"prologue": lambda: ops.encode_prologue(x), "tail": lambda: m.encoder_tail(f_tail)}
for l, f in f_glob.items():
    cases[f"global_conv[{l}]"] = (lambda f=f, l=l: (m.vn_lna_global(l, f),))
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
for name, fn in cases.items():
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    bad1 = 0
    for _ in range(R):
        out = fn()
        torch.cuda.synchronize()
        bad1 += int(not all(torch.equal(a, b) for a, b in zip(out, ref)))
    bad8, worst = 0, 0.0
    for _ in range(R):
        outs = []
        for st in streams:
            with torch.cuda.stream(st):
                outs.append(fn())
        torch.cuda.synchronize()
        for out in outs:
            ok = all(torch.equal(a, b) for a, b in zip(out, ref))
            bad8 += int(not ok)
            if not ok:
                worst = max(worst, max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(out, ref)))
    print(f"{name:18s} alone: {bad1}/{R} differ   8 streams: {bad8}/{8 * R} differ   worst relative difference {worst:.3g}")
