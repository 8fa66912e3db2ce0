"""Error of the HIP SDF decoder against an fp64 evaluation of the same decoder (dev tool; needs a GPU; the oracle is test
infrastructure, hence this lives under tests/).  Run once per GEMM mode:
    python tests/tools/sdf_accuracy.py                      # default: three-piece bf16 products
    LS_SDF_BF16X2=1 python tests/tools/sdf_accuracy.py      # opt-in two-piece mode
    LS_GEMM_MODE=fp32 python tests/tools/sdf_accuracy.py     # fp32-MFMA kernel
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from livingscenes_amd import synth
from livingscenes_amd.model_utils import Shape_Prior
from oracle import net

dev = torch.device("cuda:0")
ecfg, dcfg = synth.default_encoder_cfg(), synth.default_decoder_cfg()
dw = synth.make_decoder_weights(dcfg, 0)
sp = Shape_Prior.from_state(ecfg, dcfg, synth.make_encoder_weights(ecfg, 0), dw, device=dev)
B = 8
emb = sp.encode(synth.make_instances(B, 1024, seed=0).to(dev))
q = synth.make_queries(B, 20000, seed=3).to(dev) * emb["s"][:, None, None] + emb["t"]
sdf = sp.decoder(q, None, emb, return_sdf=True)
w64 = {k: v.double().to(dev) for k, v in net.as_params(dw).items()}
ref = net.field_query(w64, dcfg, q.double(), {k: v.double() for k, v in emb.items()})
err = (sdf.double() - ref).abs().max().item() / ref.abs().max().item()
print(f"max |sdf - fp64| / max |sdf| over {B} x 20000 queries = {err:.2e}")
