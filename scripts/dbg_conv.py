import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from fea_amd import _lib as L
DEV='cuda'
def run(x, wt, dt, bias=None):
    n, ci, h, w = x.shape; co = wt.shape[0]
    xd = x.permute(0,2,3,1).contiguous().to(dt).to(DEV)
    wp = torch.empty(9*co*ci, dtype=dt, device=DEV)
    wd = wt.to(DEV)
    L.call("fea_pack_conv3x3", wd, wp, co, ci, 0, L.dt_code(dt), L.stream_ptr())
    y = torch.full((n,h,w,co), float('nan'), dtype=dt, device=DEV)
    L.call("fea_conv3x3", xd, ci, wp, None, y, co, None, n, h, w, ci, co, L.dt_code(dt), L.stream_ptr())
    torch.cuda.synchronize()
    return y.float().cpu().permute(0,3,1,2)
def main():
    g = torch.Generator().manual_seed(0)
    for dt in (torch.float32, torch.bfloat16):
        n,ci,co,h,w = 1,64,64,32,32
        x = torch.randn(n,ci,h,w,generator=g).to(dt).float()
        for tap in range(9):
            wt = torch.zeros(co,ci,3,3); 
            for c in range(64): wt[c,c,tap//3,tap%3]=1
            y = run(x, wt, dt); ref = F.conv2d(x, wt, padding=1)
            print(dt, 'identity tap', tap, 'maxerr', (y-ref).abs().max().item(), 'nan', torch.isnan(y).sum().item())
        wt = torch.zeros(co,ci,3,3); wt[:,:,1,1] = torch.randn(co,ci,generator=g).to(dt).float()
        y = run(x, wt, dt); ref = F.conv2d(x, wt, padding=1)
        print(dt, 'random center', (y-ref).abs().max().item())
        # which output channels / pixels are wrong
        err = (y-ref).abs()
        print(' per-channel bad:', (err.amax((0,2,3))>1e-2).nonzero().flatten().tolist()[:70])
        print(' per-row bad:', (err.amax((0,1,3))>1e-2).nonzero().flatten().tolist())
        # single input channel k -> which k are wrong
        bad=[]
        for k in range(0,64):
            wt = torch.zeros(co,ci,3,3); wt[:,k,1,1]=torch.arange(1,co+1).float()
            y = run(x, wt, dt); ref = F.conv2d(x, wt, padding=1)
            if (y-ref).abs().max()>1e-2: bad.append(k)
        print(' bad k:', bad)

if __name__ == "__main__":
    main()
