import sys, time, torch
sys.path.insert(0, '.')
from livingscenes_amd import synth
from oracle import net
ecfg = synth.default_encoder_cfg(); ew = synth.make_encoder_weights(ecfg, 0)
x = synth.make_instances(4, 1024, seed=0)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    net.shape_prior_encode(ew, ecfg, x[:1])
    t0 = time.perf_counter(); net.shape_prior_encode(ew, ecfg, x[:2]); t1 = time.perf_counter()
    print(th, "threads:", 2 / (t1 - t0), "inst/s", flush=True)
