import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from scripts.dbg_conv import run
dt = torch.float32
for ci in (32, 64):
    co,h,w=64,32,32
    x = torch.zeros(1,ci,h,w)
    for c in range(ci): x[0,c]=c+1
    wt = torch.zeros(co,ci,3,3)
    for c in range(min(ci,co)): wt[c,c,1,1]=1
    y = run(x, wt, dt)
    print('ci',ci,'y[:,5,5]=', y[0,:,5,5].tolist())
    # single nonzero weight: co=3 <- ci=k
    for k in (0,1,4,5,8,31):
        wt = torch.zeros(co,ci,3,3); wt[3,k,1,1]=1
        y = run(x, wt, dt)
        nz = y[0,:,5,5].nonzero().flatten().tolist()
        print('  w[3,%d]=1 -> nonzero co %s values %s'%(k, nz, y[0,nz,5,5].tolist()))
