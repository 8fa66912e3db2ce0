import sys, torch, time
sys.path.insert(0,'/root/repo')
from livingscenes_amd import synth
from livingscenes_amd.model_utils import Shape_Prior
from oracle import net
dev=torch.device("cuda:0")
ecfg,dcfg=synth.default_encoder_cfg(),synth.default_decoder_cfg()
ew,dw=synth.make_encoder_weights(ecfg,0),synth.make_decoder_weights(dcfg,0)
sp=Shape_Prior.from_state(ecfg,dcfg,ew,dw,device=dev)
for B,N in ((2,2048),(3,512),(1,1024),(70,1024),(2,4096)):
    x=synth.make_instances(B,N,seed=B+N)
    x=x if isinstance(x,torch.Tensor) else x[0]
    emb=sp.encode(x.to(dev))
    nb=min(B,2)
    t0=time.time(); ref=net.shape_prior_encode(ew,ecfg,x[:nb]); dt=time.time()-t0
    errs={k: float((emb[k][:nb].cpu()-ref[k]).abs().max()/ref[k].abs().max()) for k in ("z_so3","z_inv","s","t")}
    print(B,N,{k:f"{v:.1e}" for k,v in errs.items()}, f"oracle {dt:.1f}s", flush=True)
