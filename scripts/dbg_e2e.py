import sys, torch
sys.path.insert(0, '.')
import fea_amd
from oracle import unet_oracle as O
from tests.helpers import rel_l2
def cancelled(k):
    p = k.split('.'); return len(p)==3 and p[1] in ('0','3') and p[2]=='bias'
for (N,H,W) in ((2,32,32),(4,64,64)):
  for dt in ('fp32','bf16'):
    torch.manual_seed(0)
    m = fea_amd.FloorPlanSegmentationModel(compute_dtype=dt).cuda().train()
    sd = {k:v.detach().cpu().clone() for k,v in m.state_dict().items()}
    params, bufs = O.split_state(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N,3,H,W,generator=g)
    dl = torch.randn(N,6,H,W,generator=g)/(N*H*W)
    taps = {}
    m._engine().debug_taps = taps
    lg = m(x.cuda())
    lg.backward(dl.cuda())
    torch.cuda.synchronize()
    taps = {k:v.cpu() for k,v in taps.items()}
    for mode in ('free','forced'):
        forced = O.forced_from_taps(taps) if mode=='forced' else None
        lo, go, nb = O.grads_from_dlogits(params, bufs, x, dl, forced=forced)
        errs = sorted(((rel_l2(pp.grad, go[k]),k) for k,pp in m.named_parameters() if not cancelled(k)), reverse=True)
        print(N,H,W,dt,mode,'logits %.2e'%rel_l2(lg,lo),'worst grads', ['%s %.2e'%(k,e) for e,k in errs[:3]], 'median %.2e'%errs[len(errs)//2][0])
