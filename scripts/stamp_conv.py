import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = torch.zeros(16, dtype=torch.int64, device="cuda")
os.environ["FEA_STAMPS_PTR"] = str(st.data_ptr())
from fea_amd import _lib as L
dt = torch.bfloat16; code = L.dt_code(dt)
for (n, h, w, ci, co) in [(64, 512, 512, 64, 64), (64, 512, 512, 128, 64), (64, 256, 256, 128, 128), (64, 64, 64, 512, 512)]:
    x = torch.randn(n, h, w, ci, device="cuda").to(dt)
    wp = torch.randn(9 * ci * co, device="cuda").to(dt)
    y = torch.empty(n, h, w, co, dtype=dt, device="cuda")
    nt = L.call("fea_conv_num_tiles", n, h, w)
    stats = torch.empty(nt * co * 2, device="cuda")
    bias = torch.zeros(co, device="cuda")
    for it in range(3):
        st.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("fea_conv3x3", x, ci, wp, bias, y, co, stats, n, h, w, ci, co, code, L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
    v = st.cpu().tolist()
    k = max(v[8], 1)
    names = ["setup", "issue A", "wait A", "taps", "bias+stats", "cvt+stage", "barrier", "store"]
    print(f"{ci}->{co}: {e0.elapsed_time(e1):.3f} ms; cycles per sampled WG (wave 0): " + ", ".join(f"{n} {v[i]/k:.0f}" for i, n in enumerate(names)))
