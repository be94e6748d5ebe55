import sys, torch
sys.path.insert(0, '.')
import fea_amd
from oracle import unet_oracle as O
from tests.helpers import rel_l2
torch.manual_seed(0)
N,H,W = 2,32,32
for dt in ('fp32','bf16'):
    torch.manual_seed(0)
    m = fea_amd.FloorPlanSegmentationModel(compute_dtype=dt).cuda().train()
    sd = {k:v.detach().cpu().clone() for k,v in m.state_dict().items()}
    params, bufs = O.split_state(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N,3,H,W,generator=g)
    dl = torch.randn(N,6,H,W,generator=g)/(N*H*W)
    taps_o = {}
    p = {k:v.clone().requires_grad_(True) for k,v in params.items()}
    lo, nb = O.forward(p, bufs, x, True, taps_o)
    go = torch.autograd.grad(lo, list(p.values()), grad_outputs=dl)
    go = dict(zip(p.keys(), go))
    taps = {}
    m._engine().debug_taps = taps
    lg = m(x.cuda())
    lg.backward(dl.cuda())
    torch.cuda.synchronize()
    print('==', dt, 'logits rel_l2', rel_l2(lg, lo))
    for k in taps_o:
        if k in taps: print('  %-20s %.3e'%(k, rel_l2(taps[k], taps_o[k])))
    for k,pp in m.named_parameters():
        print('  grad %-22s %.3e'%(k, rel_l2(pp.grad, go[k])))
    for k,b in m.named_buffers():
        if 'num_batches' in k: continue
        e = rel_l2(b, nb[k])
        if e>1e-5: print('  buf', k, e)
