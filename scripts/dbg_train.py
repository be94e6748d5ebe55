import sys, torch
sys.path.insert(0, '.')
import fea_amd
from fea_amd.train import UNetTrainer
from fea_amd import functional as Fn
from oracle import unet_oracle as O
from tests.helpers import rel_l2
torch.manual_seed(12)
m = fea_amd.FloorPlanSegmentationModel(compute_dtype='fp32').cuda()
params, bufs = O.split_state({k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
tr = UNetTrainer(m)
x, tgt = O.synthetic_batch(2, 32, 32, seed=99)
g = torch.Generator().manual_seed(3)
x = x + 0.1 * torch.randn(x.shape, generator=g)
state = {}
v0 = m.enc1[0].weight._version
loss = tr.step(x.cuda(), tgt.cuda())
print('version bump', v0, m.enc1[0].weight._version)
lo, _, grads, nb = O.loss_and_grads(params, bufs, x, tgt, training=True)
print('step0 loss', loss.item(), lo.item())
errs = sorted(((rel_l2(p.grad, grads[k]), k) for k, p in m.named_parameters()), reverse=True)
print('grad errs (pre-clip vs post-clip scale differs) top', errs[:4])
clipped, norm = O.clip_grad_norm(grads, 1.0)
print('norm', tr.last_grad_norm.item(), norm.item())
params = O.adamw_step(params, clipped, state); bufs = nb
errs = sorted(((rel_l2(p.detach(), params[k]), k) for k, p in m.named_parameters()), reverse=True)
print('param errs top', errs[:6])
# loss at new params: same model (cached packs) vs fresh model with same state
m.train()
l_same = Fn.bce_dice_loss(m(x.cuda()), tgt.cuda()).item()
m2 = fea_amd.FloorPlanSegmentationModel(compute_dtype='fp32').cuda(); m2.load_state_dict(m.state_dict()); m2.train()
l_fresh = Fn.bce_dice_loss(m2(x.cuda()), tgt.cuda()).item()
lo2 = O.bce_dice_loss(O.forward(params, bufs, x, True)[0], tgt).item()
p_hip, b_hip = O.split_state({k: v.detach().cpu().clone() for k, v in m2.state_dict().items()})
lo3 = O.bce_dice_loss(O.forward(p_hip, b_hip, x, True)[0], tgt).item()
print('loss same-model', l_same, 'fresh-model', l_fresh, 'oracle(own params)', lo2, 'oracle(hip params)', lo3)
