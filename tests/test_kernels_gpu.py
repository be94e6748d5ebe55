"""Kernel-level parity: every C-ABI compute entry point against a plain PyTorch
fp32 CPU evaluation of the same reference layer (nn.Conv2d / BatchNorm2d /
MaxPool2d / ConvTranspose2d semantics, models.py:15-34) on the same inputs.

Tolerances (stated, per north_star): fp32 path <= 1e-4 relative (we assert
1e-5 L2 / 1e-4 peak); bf16 path: inputs are rounded to bf16 on BOTH sides, so
the difference is the kernel's output rounding (2^-9) and fp32 accumulation
order -> 4e-3 L2-relative, 2e-2 peak-relative.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from fea_amd import _lib as L
from tests.helpers import max_rel, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return (1e-5, 1e-4) if dt == torch.float32 else (4e-3, 2e-2)


def check(got, ref, dt, what="", scale=1.0):
    l2, pk = tol(dt)
    e2, ep = rel_l2(got.float().cpu(), ref), max_rel(got.float().cpu(), ref)
    assert e2 < l2 * scale and ep < pk * scale, f"{what}: rel_l2={e2:.3e} peak={ep:.3e} ({dt})"


def rnd(dt, *shape, gen):
    """Random tensor representable in dt (so both sides see identical inputs)."""
    return torch.randn(*shape, generator=gen).to(dt).float()


def to_nhwc(x, dt, ld=None, off=0):
    """NCHW fp32 CPU -> device NHWC tensor of dtype dt; optionally embedded at
    channel offset `off` of a wider (ld) buffer filled with NaN canaries."""
    n, c, h, w = x.shape
    ld = ld or c
    buf = torch.full((n, h, w, ld), float("nan"), dtype=dt, device=DEV)
    buf[..., off:off + c] = x.permute(0, 2, 3, 1).to(dt).to(DEV)
    return buf


def view_ptr(buf, off):
    return buf.data_ptr() + off * buf.element_size()


def from_nhwc(buf, c, off=0):
    return buf[..., off:off + c].float().cpu().permute(0, 3, 1, 2).contiguous()


def S():
    return L.stream_ptr()


SHAPES_3X3 = [  # N, H, W, Cin, Cout
    (2, 32, 32, 64, 64),
    (1, 16, 16, 128, 128),      # TW = 16 tiles
    (1, 24, 40, 64, 128),       # partial tiles in both directions
    (1, 8, 8, 192, 64),         # tile larger than the image
    (3, 16, 64, 64, 192),       # BN = 64 path with several cout tiles
    (1, 8, 32, 1024, 128),      # deep K (16 chunks) on the 8x32-tile, 128-column kernel (bottleneck / dec4.0 shapes)
    (2, 16, 32, 512, 1024),     # the same with 8 column tiles and two tile rows
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", SHAPES_3X3)
def test_conv3x3_fwd_stats_and_dgrad(dt, shape):
    n, h, w, ci, co = shape
    g = torch.Generator().manual_seed(1)
    x = rnd(dt, n, ci, h, w, gen=g)
    wt = rnd(dt, co, ci, 3, 3, gen=g) / (3 * ci ** 0.5)
    wt = wt.to(dt).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)

    xd = to_nhwc(x, dt)
    wp = torch.empty(9 * co * ci, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3", wt.to(DEV), wp, co, ci, 0, L.dt_code(dt), S())
    ldy, off = co + 64, 64                      # write into a channel slice
    y = torch.full((n, h, w, ldy), float("nan"), dtype=dt, device=DEV)
    ntiles = L.call("fea_conv_num_tiles", n, h, w)
    stats = torch.full((ntiles, co, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3", xd, ci, wp, bias.to(DEV), view_ptr(y, off), ldy, stats, n, h, w, ci, co, L.dt_code(dt), S())
    torch.cuda.synchronize()
    check(from_nhwc(y, co, off), ref, dt, "conv3x3 fwd")
    assert torch.isnan(y[..., :off].float()).all(), "wrote outside its channel slice"
    s = stats.double().sum(0).cpu()
    ref_s = torch.stack([ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))], 1)
    assert rel_l2(s, ref_s) < 1e-5, "BN partial sums"

    # dgrad: same kernel, dgrad-packed weights, no bias
    dy = rnd(dt, n, co, h, w, gen=g)
    ref_dx = F.conv_transpose2d(dy, wt, padding=1)
    wpd = torch.empty(9 * co * ci, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3", wt.to(DEV), wpd, co, ci, 1, L.dt_code(dt), S())
    dx = torch.full((n, h, w, ci), float("nan"), dtype=dt, device=DEV)
    L.call("fea_conv3x3", to_nhwc(dy, dt), co, wpd, None, dx, ci, None, n, h, w, co, ci, L.dt_code(dt), S())
    torch.cuda.synchronize()
    check(from_nhwc(dx, ci), ref_dx, dt, "conv3x3 dgrad")


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64), (1, 24, 40, 128, 128), (1, 16, 16, 128, 256), (2, 64, 64, 128, 64)])
def test_conv3x3_dgrad_with_fused_bn_backward_reduction(shape):
    """fea_conv3x3_dgrad_bnred == fea_conv3x3 (dgrad pack) followed by fea_bn_relu_bwd_reduce:
    identical da (bit for bit) and the same per-channel sums."""
    n, h, w, ck, cn = shape                      # ck = channels of dy (GEMM K), cn = channels of da / y
    dt = torch.bfloat16
    code = L.dt_code(dt)
    g = torch.Generator().manual_seed(11)
    dy = rnd(dt, n, ck, h, w, gen=g)
    wt = (rnd(dt, ck, cn, 3, 3, gen=g) / (3 * ck ** 0.5)).to(dt).float()      # Conv2d(cn -> ck).weight
    y = rnd(dt, n, cn, h, w, gen=g)
    scale, shift = torch.randn(cn, generator=g), torch.randn(cn, generator=g)
    mean, invstd = torch.randn(cn, generator=g), torch.rand(cn, generator=g) + 0.5
    dyd, yd = to_nhwc(dy, dt), to_nhwc(y, dt)
    wpd = torch.empty(9 * ck * cn, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3", wt.to(DEV), wpd, ck, cn, 1, code, S())
    coef = [t.to(DEV) for t in (scale, shift, mean, invstd)]
    da_ref = torch.full((n, h, w, cn), float("nan"), dtype=dt, device=DEV)
    L.call("fea_conv3x3", dyd, ck, wpd, None, da_ref, cn, None, n, h, w, ck, cn, code, S())
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, cn)
    part_ref = torch.zeros(nblk, cn, 2, device=DEV)
    L.call("fea_bn_relu_bwd_reduce", da_ref, cn, None, yd, *coef, part_ref, n, h, w, cn, code, S())
    ntiles = L.call("fea_conv_num_tiles", n, h, w)
    da = torch.full((n, h, w, cn), float("nan"), dtype=dt, device=DEV)
    part = torch.full((ntiles, cn, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3_dgrad_bnred", dyd, ck, wpd, da, cn, yd, cn, *coef, part, n, h, w, ck, cn, code, S())
    torch.cuda.synchronize()
    assert torch.equal(da.view(torch.int16), da_ref.view(torch.int16)), "fused kernel changed the data gradient"
    got, ref = part.double().sum(0).cpu(), part_ref.double().sum(0).cpu()
    assert rel_l2(got, ref) < 1e-5, f"fused BN-backward sums: {rel_l2(got, ref):.3e}"


SHAPES_P = [  # N, H, W, Cin, Cout -- shapes the persistent kernels serve (H % 8 == 0, W % 32 == 0, channels 64 / 128)
    (2, 32, 32, 64, 64),
    (1, 8, 32, 64, 64),         # a single tile: every halo line is an image border
    (3, 24, 96, 64, 64),        # odd tile counts: workgroups with different numbers of tiles
    (2, 16, 64, 128, 64),       # two K chunks
    (2, 16, 64, 64, 128),       # two column groups
    (1, 40, 64, 128, 128),
    (2, 16, 96, 128, 256),      # four column groups
    (5, 64, 128, 64, 64),       # 320 tiles: more tiles than workgroups, several tiles per workgroup
]


@pytest.mark.parametrize("shape", SHAPES_P)
def test_persistent_conv3x3_fwd_stats_dgrad_and_bnred(shape):
    """fea_conv3x3p / fea_conv3x3p_dgrad_bnred (conv_p.hip) against F.conv2d on bf16-rounded inputs and against the
    plain entry points they replace on the shallow levels: same outputs to bf16 rounding of the same fp32 sums, same
    BatchNorm partial sums, same fused BatchNorm-backward sums."""
    n, h, w, ci, co = shape
    dt = torch.bfloat16
    code = L.dt_code(dt)
    rows = L.call("fea_conv3x3p_rows", n, h, w, ci, co, code)
    assert rows > 0
    g = torch.Generator().manual_seed(31)
    x = rnd(dt, n, ci, h, w, gen=g)
    wt = (rnd(dt, co, ci, 3, 3, gen=g) / (3 * ci ** 0.5)).to(dt).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    xd = to_nhwc(x, dt, ld=ci + 64, off=64)
    wp = torch.empty(9 * co * ci, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3_frag", wt.to(DEV), wp, co, ci, 0, code, S())
    ldy, off = co + 64, 64
    y = torch.full((n, h, w, ldy), float("nan"), dtype=dt, device=DEV)
    stats = torch.full((rows, co, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3p", view_ptr(xd, 64), ci + 64, wp, bias.to(DEV), view_ptr(y, off), ldy, stats, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(y, co, off), ref, dt, "persistent conv3x3 fwd")
    assert torch.isnan(y[..., :off].float()).all(), "wrote outside its channel slice"
    s = stats.double().sum(0).cpu()
    ref_s = torch.stack([ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))], 1)
    assert rel_l2(s, ref_s) < 1e-5, "BN partial sums"
    # without bias / stats (plain data gradient), against the plain kernel bit for bit is not required (different
    # summation order inside the MFMA chain is the same here: same k order) -- compare to the reference
    dy = rnd(dt, n, co, h, w, gen=g)
    ref_dx = F.conv_transpose2d(dy, wt, padding=1)
    wpd = torch.empty(9 * co * ci, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3_frag", wt.to(DEV), wpd, co, ci, 1, code, S())
    dyd = to_nhwc(dy, dt)
    dx = torch.full((n, h, w, ci), float("nan"), dtype=dt, device=DEV)
    rows_d = L.call("fea_conv3x3p_rows", n, h, w, co, ci, code)
    if rows_d == 0:
        return                                    # the transposed shape (K = 256 channels) is not served
    L.call("fea_conv3x3p", dyd, co, wpd, None, dx, ci, None, n, h, w, co, ci, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(dx, ci), ref_dx, dt, "persistent conv3x3 dgrad")
    # fused BatchNorm-backward reduction: da identical to the plain dgrad above, sums equal to fea_bn_relu_bwd_reduce's
    yb = rnd(dt, n, ci, h, w, gen=g)
    scale, shift = torch.randn(ci, generator=g), torch.randn(ci, generator=g)
    mean, invstd = torch.randn(ci, generator=g), torch.rand(ci, generator=g) + 0.5
    coef = [t.to(DEV) for t in (scale, shift, mean, invstd)]
    ybd = to_nhwc(yb, dt)
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, ci)
    part_ref = torch.zeros(nblk, ci, 2, device=DEV)
    L.call("fea_bn_relu_bwd_reduce", dx, ci, None, ybd, *coef, part_ref, n, h, w, ci, code, S())
    da = torch.full((n, h, w, ci), float("nan"), dtype=dt, device=DEV)
    part = torch.full((rows_d, ci, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3p_dgrad_bnred", dyd, co, wpd, da, ci, ybd, ci, *coef, part, n, h, w, co, ci, code, S())
    torch.cuda.synchronize()
    assert torch.equal(da.view(torch.int16), dx.view(torch.int16)), "fused kernel changed the data gradient"
    got, want = part.double().sum(0).cpu(), part_ref.double().sum(0).cpu()
    assert rel_l2(got[:, 0], want[:, 0]) < 1e-5 and rel_l2(got[:, 1], want[:, 1]) < 1e-4, (rel_l2(got[:, 0], want[:, 0]), rel_l2(got[:, 1], want[:, 1]))


@pytest.mark.parametrize("shape", SHAPES_P)
@pytest.mark.parametrize("pooled", [False, True])
def test_persistent_conv3x3_bias_relu_pool_inference(shape, pooled):
    """fea_conv3x3p_bias_relu (folded-BatchNorm inference, persistent form) against F.conv2d / relu / max_pool2d on
    bf16-rounded inputs; the pooled output must be exactly the max over the stored (bf16) activations."""
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, h, w, ci, co = shape
    g = torch.Generator().manual_seed(33)
    x = rnd(dt, n, ci, h, w, gen=g)
    wt = (rnd(dt, co, ci, 3, 3, gen=g) / (3 * ci ** 0.5)).to(dt).float()
    bias = torch.randn(co, generator=g) * 0.3
    ref = F.relu(F.conv2d(x, wt, bias, padding=1))
    wp = torch.empty(9 * co * ci, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3_frag", wt.to(DEV), wp, co, ci, 0, code, S())
    ldy, off = co + 64, 64
    y = torch.full((n, h, w, ldy), float("nan"), dtype=dt, device=DEV)
    pool = torch.full((n, h // 2, w // 2, co + 8), float("nan"), dtype=dt, device=DEV) if pooled else None
    L.call("fea_conv3x3p_bias_relu", to_nhwc(x, dt), ci, wp, bias.to(DEV), view_ptr(y, off), ldy,
           view_ptr(pool, 8) if pooled else None, co + 8 if pooled else 0, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    got = from_nhwc(y, co, off)
    check(got, ref, dt, "persistent conv+bias+relu")
    assert (got >= 0).all() and torch.isnan(y[..., :off].float()).all()
    if pooled:
        assert torch.equal(from_nhwc(pool, co, 8), F.max_pool2d(got, 2)), "pooled output != max of the stored activations"
        assert torch.isnan(pool[..., :8].float()).all()


def test_persistent_conv3x3_rejects_shapes_it_does_not_serve():
    code = L.dt_code(torch.bfloat16)
    assert L.call("fea_conv3x3p_rows", 1, 24, 40, 64, 64, code) == 0          # W % 32
    assert L.call("fea_conv3x3p_rows", 1, 32, 32, 256, 128, code) == 0        # Cin
    assert L.call("fea_conv3x3p_rows", 1, 32, 32, 64, 512, code) == 0         # Cout
    assert L.call("fea_conv3x3p_rows", 1, 32, 32, 64, 64, L.F32) == 0         # fp32
    x = torch.zeros(64, device=DEV)
    with pytest.raises(L.FeaError):
        L.call("fea_conv3x3p", x, 64, x, None, x, 64, None, 1, 24, 40, 64, 64, code, S())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64), (1, 16, 16, 128, 64), (1, 24, 40, 64, 128), (1, 8, 8, 64, 192)])
def test_conv3x3_wgrad(dt, shape):
    n, h, w, ci, co = shape
    g = torch.Generator().manual_seed(2)
    x = rnd(dt, n, ci, h, w, gen=g)
    dy = rnd(dt, n, co, h, w, gen=g)
    xr = x.clone().requires_grad_(False)
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(xr, wt, padding=1).backward(dy)
    ref = wt.grad
    code = L.dt_code(dt)
    nbytes = L.load().fea_wgrad_ws_bytes(0, n, h, w, co, ci, code)
    ws = torch.empty(nbytes // 4, device=DEV)
    dW = torch.full((co, ci, 3, 3), float("nan"), device=DEV)
    # inputs live in channel slices of wider buffers
    xd, dyd = to_nhwc(x, dt, ld=ci + 64, off=64), to_nhwc(dy, dt, ld=co + 128, off=128)
    L.call("fea_conv3x3_wgrad", view_ptr(dyd, 128), co + 128, view_ptr(xd, 64), ci + 64, dW, ws, nbytes,
           n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    check(dW, ref, dt, "conv3x3 wgrad", scale=0.5 if dt == torch.bfloat16 else 1.0)


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64), (1, 16, 16, 128, 128), (1, 24, 40, 64, 128), (1, 8, 8, 192, 64),
                                   (3, 16, 64, 64, 192), (1, 2, 2, 128, 256)])
@pytest.mark.parametrize("pooled", [False, True])
def test_conv3x3_bias_relu_pool_inference_kernel(shape, pooled):
    """Folded-BatchNorm inference kernel: relu(conv3x3 + bias) (+ MaxPool2d(2)) against
    F.conv2d / F.relu / F.max_pool2d on bf16-rounded inputs.  The pooled output must be exactly
    the max over the stored (bf16) activations."""
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, h, w, ci, co = shape
    g = torch.Generator().manual_seed(21)
    x = rnd(dt, n, ci, h, w, gen=g)
    wt = (rnd(dt, co, ci, 3, 3, gen=g) / (3 * ci ** 0.5)).to(dt).float()
    bias = torch.randn(co, generator=g) * 0.3
    ref = F.relu(F.conv2d(x, wt, bias, padding=1))
    wp = torch.empty(9 * co * ci, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3", wt.to(DEV), wp, co, ci, 0, code, S())
    ldy, off = co + 64, 64
    y = torch.full((n, h, w, ldy), float("nan"), dtype=dt, device=DEV)
    pool = torch.full((n, h // 2, w // 2, co + 8), float("nan"), dtype=dt, device=DEV) if pooled else None
    L.call("fea_conv3x3_bias_relu", to_nhwc(x, dt), ci, wp, bias.to(DEV), view_ptr(y, off), ldy,
           view_ptr(pool, 8) if pooled else None, co + 8 if pooled else 0, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    got = from_nhwc(y, co, off)
    check(got, ref, dt, "conv+bias+relu")
    assert (got >= 0).all() and torch.isnan(y[..., :off].float()).all()
    if pooled:
        assert torch.equal(from_nhwc(pool, co, 8), F.max_pool2d(got, 2)), "pooled output != max of the stored activations"
        assert torch.isnan(pool[..., :8].float()).all()


@pytest.mark.parametrize("cin", [3, 1, 5, 7])
@pytest.mark.parametrize("hw", [(32, 48), (16, 16), (40, 72), (32, 64), (8, 96)])      # the last two: whole 8x32 tiles, the persistent kernel
def test_first_layer_direct_path(cin, hw):
    """enc1.0 = Conv2d(in_channels, 64, 3, padding=1) straight from the fp32 NCHW input (bf16
    path): forward + BN partial sums + weight gradient against F.conv2d on bf16-rounded inputs."""
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, (h, w), co = 2, hw, 64
    g = torch.Generator().manual_seed(13)
    x = rnd(dt, n, cin, h, w, gen=g)                      # representable in bf16: both sides see the same values
    wt = (rnd(dt, co, cin, 3, 3, gen=g) / (3 * cin ** 0.5)).to(dt).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    ks = (9 * cin + 31) // 32
    wp = torch.empty(ks * co * 32, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3_first", wt.to(DEV), wp, co, cin, code, S())
    rows_p = L.call("fea_conv3x3_first_rows", n, cin, h, w, co, code)
    assert (rows_p > 0) == (h % 8 == 0 and w % 32 == 0)
    ntiles = rows_p or L.call("fea_conv_num_tiles", n, h, w)
    stats = torch.full((ntiles, co, 2), float("nan"), device=DEV)
    y = torch.full((n, h, w, co + 64), float("nan"), dtype=dt, device=DEV)
    xd = x.to(DEV).contiguous()
    L.call("fea_conv3x3_first", xd, wp, bias.to(DEV), view_ptr(y, 64), co + 64, stats, n, cin, h, w, co, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(y, co, 64), ref, dt, "first-layer conv")
    assert torch.isnan(y[..., :64].float()).all(), "wrote outside its channel slice"
    y2 = torch.full((n, h, w, co), float("nan"), dtype=dt, device=DEV)
    L.call("fea_conv3x3_first_bias_relu", xd, wp, bias.to(DEV), y2, co, n, cin, h, w, co, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(y2, co), F.relu(ref), dt, "first-layer conv + bias + relu (inference)")
    s = stats.double().sum(0).cpu()
    ref_s = torch.stack([ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))], 1)
    assert rel_l2(s, ref_s) < 1e-5, "BN partial sums"
    # weight gradient
    dy = rnd(dt, n, co, h, w, gen=g)
    wz = torch.zeros(co, cin, 3, 3, requires_grad=True)
    F.conv2d(x, wz, padding=1).backward(dy)
    nbytes = L.load().fea_first_wgrad_ws_bytes(n, h, w, co)
    ws = torch.empty(nbytes // 4, device=DEV)
    dW = torch.full((co, cin, 3, 3), float("nan"), device=DEV)
    dyd = to_nhwc(dy, dt, ld=co + 128, off=128)
    L.call("fea_conv3x3_first_wgrad", view_ptr(dyd, 128), co + 128, xd, dW, ws, nbytes, n, cin, h, w, co, code, S())
    torch.cuda.synchronize()
    check(dW, wz.grad, dt, "first-layer wgrad", scale=0.5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin", [3, 1, 5])
def test_enc1_im2col_path(dt, cin):
    """enc1.0 = Conv2d(in_channels, 64, 3, padding=1) as im2col rows + 1x1 GEMM."""
    n, h, w, co, kpad = 2, 32, 48, 64, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(co, cin, 3, 3, generator=g) / 5
    bias = torch.randn(co, generator=g)
    code = L.dt_code(dt)
    xcol = torch.full((n, h, w, kpad), float("nan"), dtype=dt, device=DEV)
    L.call("fea_im2col3x3_nchw", x.to(DEV), xcol, n, cin, h, w, kpad, code, S())
    wp = torch.empty(co * kpad, dtype=dt, device=DEV)
    L.call("fea_pack_conv3x3_im2col", wt.to(DEV), wp, co, cin, kpad, code, S())
    y = torch.empty((n, h, w, co), dtype=dt, device=DEV)
    ntiles = L.call("fea_conv_num_tiles", n, h, w)
    stats = torch.empty((ntiles, co, 2), device=DEV)
    L.call("fea_conv1x1", xcol, kpad, wp, bias.to(DEV), y, co, stats, n, h, w, kpad, co, code, S())
    torch.cuda.synchronize()
    xq, wq = x.to(dt).float(), wt.to(dt).float()
    ref = F.conv2d(xq, wq, bias, padding=1)
    check(from_nhwc(y, co), ref, dt, "enc1.0 fwd")
    # weight gradient through the same rows
    dy = rnd(dt, n, co, h, w, gen=g)
    wz = torch.zeros(co, cin, 3, 3, requires_grad=True)
    F.conv2d(xq, wz, padding=1).backward(dy)
    nbytes = L.load().fea_wgrad_ws_bytes(1, n, h, w, co, kpad, code)
    ws = torch.empty(nbytes // 4, device=DEV)
    dW = torch.full((co, cin, 3, 3), float("nan"), device=DEV)
    L.call("fea_conv1x1_im2col_wgrad", to_nhwc(dy, dt), co, xcol, kpad, dW, ws, nbytes, n, h, w, kpad, cin, co, code, S())
    torch.cuda.synchronize()
    check(dW, wz.grad, dt, "enc1.0 wgrad")


@pytest.mark.parametrize("shape", [(2, 16, 16, 128, 64), (1, 8, 24, 256, 128), (1, 2, 2, 1024, 512), (2, 32, 32, 64, 64), (1, 40, 24, 128, 64)])
def test_upconv2x2_bwd_data_with_fused_bn_backward_reduction(shape):
    """fea_upconv2x2_bwd_data_bnred == fea_upconv2x2_bwd_data followed by fea_bn_relu_bwd_reduce:
    identical dx (bit for bit) and the same per-channel sums."""
    n, h, w, ci, co = shape
    dt = torch.bfloat16
    code = L.dt_code(dt)
    g = torch.Generator().manual_seed(17)
    wt = (torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5).to(dt).float()
    dy = rnd(dt, n, co, 2 * h, 2 * w, gen=g)
    y = rnd(dt, n, ci, h, w, gen=g)
    scale, shift = torch.randn(ci, generator=g), torch.randn(ci, generator=g)
    mean, invstd = torch.randn(ci, generator=g), torch.rand(ci, generator=g) + 0.5
    coef = [t.to(DEV) for t in (scale, shift, mean, invstd)]
    dyd = to_nhwc(dy, dt, ld=2 * co, off=0)
    yd = to_nhwc(y, dt)
    wpb = torch.empty(4 * ci * co, dtype=dt, device=DEV)
    L.call("fea_pack_upconv2x2", wt.to(DEV), wpb, ci, co, 1, code, S())
    dx_ref = torch.full((n, h, w, ci), float("nan"), dtype=dt, device=DEV)
    L.call("fea_upconv2x2_bwd_data", dyd, 2 * co, wpb, dx_ref, ci, n, h, w, ci, co, code, S())
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, ci)
    part_ref = torch.zeros(nblk, ci, 2, device=DEV)
    L.call("fea_bn_relu_bwd_reduce", dx_ref, ci, None, yd, *coef, part_ref, n, h, w, ci, code, S())
    ntiles = L.call("fea_conv_num_tiles", n, h, w)
    dx = torch.full((n, h, w, ci), float("nan"), dtype=dt, device=DEV)
    part = torch.full((ntiles, ci, 2), float("nan"), device=DEV)
    L.call("fea_upconv2x2_bwd_data_bnred", dyd, 2 * co, wpb, dx, ci, yd, ci, *coef, part, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)), "fused kernel changed the data gradient"
    got, ref = part.double().sum(0).cpu(), part_ref.double().sum(0).cpu()
    assert rel_l2(got, ref) < 1e-5, f"fused BN-backward sums: {rel_l2(got, ref):.3e}"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 128, 64), (1, 8, 24, 256, 128), (1, 2, 2, 1024, 512), (2, 32, 32, 64, 64)])
def test_upconv2x2(dt, shape):
    """ConvTranspose2d(ci, co, 2, stride=2): forward into the concat buffer, data
    gradient, weight gradient, bias gradient."""
    n, h, w, ci, co = shape
    g = torch.Generator().manual_seed(4)
    code = L.dt_code(dt)
    x = rnd(dt, n, ci, h, w, gen=g)
    wt = (torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5).to(dt).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv_transpose2d(x, wt, bias, stride=2)
    wp = torch.empty(4 * ci * co, dtype=dt, device=DEV)
    L.call("fea_pack_upconv2x2", wt.to(DEV), wp, ci, co, 0, code, S())
    cat = torch.full((n, 2 * h, 2 * w, 2 * co), float("nan"), dtype=dt, device=DEV)
    L.call("fea_upconv2x2_fwd", to_nhwc(x, dt), ci, wp, bias.to(DEV), cat, 2 * co, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(cat, co, 0), ref, dt, "upconv fwd")
    assert torch.isnan(cat[..., co:].float()).all(), "skip half of the concat buffer was touched"

    dy = rnd(dt, n, co, 2 * h, 2 * w, gen=g)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    F.conv_transpose2d(xr, wr, br, stride=2).backward(dy)
    dyd = to_nhwc(dy, dt, ld=2 * co, off=0)     # gradient of the concat buffer's first half
    wpb = torch.empty(4 * ci * co, dtype=dt, device=DEV)
    L.call("fea_pack_upconv2x2", wt.to(DEV), wpb, ci, co, 1, code, S())
    dx = torch.full((n, h, w, ci), float("nan"), dtype=dt, device=DEV)
    L.call("fea_upconv2x2_bwd_data", dyd, 2 * co, wpb, dx, ci, n, h, w, ci, co, code, S())
    nbytes = L.load().fea_wgrad_ws_bytes(2, n, h, w, ci, co, code)
    ws = torch.empty(nbytes // 4, device=DEV)
    dW = torch.full((ci, co, 2, 2), float("nan"), device=DEV)
    L.call("fea_upconv2x2_wgrad", to_nhwc(x, dt), ci, dyd, 2 * co, dW, ws, nbytes, n, h, w, ci, co, code, S())
    part = torch.empty((2048, co, 2), device=DEV)
    nb = ctypes.c_int(0)
    L.call("fea_channel_sum", dyd, 2 * co, n * 4 * h * w, co, part, ctypes.byref(nb), code, S())
    db = torch.empty(co, device=DEV)
    L.call("fea_reduce_partials", part, nb.value, co, 0, db, S())
    torch.cuda.synchronize()
    check(from_nhwc(dx, ci), xr.grad, dt, "upconv bwd data")
    check(dW, wr.grad, dt, "upconv wgrad")
    assert rel_l2(db.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 32, 64, True), (1, 8, 8, 1024, False), (3, 32, 32, 128, True), (1, 2, 2, 256, False)])
def test_bn_relu_pool_forward_backward(dt, shape):
    """conv output -> BatchNorm2d(train) -> ReLU (-> MaxPool2d(2)) and its
    backward, against autograd on the same ops.  The conv output is kept away
    from the ReLU kink and from pool ties so the comparison is smooth."""
    n, h, w, c, pool = shape
    g = torch.Generator().manual_seed(5)
    code = L.dt_code(dt)
    y = (rnd(dt, n, c, h, w, gen=g) * 2 + 0.3).to(dt).float()
    gamma = torch.rand(c, generator=g) + 0.5
    gamma[::7] *= -1                                   # negative gamma must work too
    beta = torch.randn(c, generator=g) * 0.2
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    bn = F.batch_norm(yr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5)
    # keep |bn| away from 0 where the two sides could disagree on the mask
    keep = bn.detach().abs() > 2e-2
    a = torch.relu(bn)
    outs = [a]
    if pool:
        outs.append(F.max_pool2d(a, 2))

    # device side: partial stats as the conv epilogue would have produced them (one "tile")
    yd = to_nhwc(y, dt)
    stats = torch.stack([y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))], 1).float().to(DEV).reshape(1, c, 2).contiguous()
    scale, shift, smean, sinv = (torch.empty(c, device=DEV) for _ in range(4))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    L.call("fea_bn_finalize", stats, 1, n * h * w, gamma.to(DEV), beta.to(DEV), rmd, rvd, 0.1, 1e-5, scale, shift, smean, sinv, c, S())
    ld = 2 * c
    full = torch.full((n, h, w, ld), float("nan"), dtype=dt, device=DEV)
    pooled = torch.full((n, h // 2, w // 2, c), float("nan"), dtype=dt, device=DEV) if pool else None
    L.call("fea_bn_relu_pool", yd, scale, shift, view_ptr(full, c), ld, pooled, n, h, w, c, code, S())
    torch.cuda.synchronize()
    assert rel_l2(rmd.cpu(), rm_ref) < 1e-5 and rel_l2(rvd.cpu(), rv_ref) < 1e-5, "running stats"
    check(from_nhwc(full, c, c), a.detach(), dt, "bn+relu")
    if pool:
        check(from_nhwc(pooled, c), outs[1].detach(), dt, "maxpool")

    # backward
    da = rnd(dt, n, c, h, w, gen=g) * keep
    dp = rnd(dt, n, c, h // 2, w // 2, gen=g) if pool else None
    loss = (a * da).sum() + ((outs[1] * dp).sum() if pool else 0)
    loss.backward()
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, c)
    part = torch.full((nblk, c, 2), float("nan"), device=DEV)
    dad = to_nhwc(da, dt, ld=ld, off=c)
    dpd = to_nhwc(dp, dt) if pool else None
    L.call("fea_bn_relu_bwd_reduce", view_ptr(dad, c), ld, dpd, yd, scale, shift, smean, sinv, part, n, h, w, c, code, S())
    dgamma, dbeta, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(c, 2, device=DEV)
    L.call("fea_bn_bwd_finalize", part, nblk, n * h * w, gamma.to(DEV), sinv, dgamma, dbeta, coef, c, S())
    dy = torch.full((n, h, w, c), float("nan"), dtype=dt, device=DEV)
    L.call("fea_bn_relu_bwd_apply", view_ptr(dad, c), ld, dpd, yd, scale, shift, smean, sinv, coef, dy, part, n, h, w, c, code, S())
    dbias = torch.empty(c, device=DEV)
    L.call("fea_reduce_partials", part, nblk, c, 0, dbias, S())
    torch.cuda.synchronize()
    assert rel_l2(dgamma.cpu(), gr.grad) < 2e-4 and rel_l2(dbeta.cpu(), br.grad) < 2e-4
    # (gaussian inputs: pool windows have no exact ties; exact ties are covered by
    # test_maxpool_backward_exact_ties_go_to_the_first_maximum)
    check(from_nhwc(dy, c), yr.grad, dt, "bn+relu(+pool) backward", scale=3.0)
    # conv-bias gradient under BN is rounding noise around 0
    assert dbias.abs().max().item() < 1e-2 * (yr.grad.abs().sum((0, 2, 3)).max().item() + 1e-6)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ncls", [6, 1, 11])
def test_head_forward_backward(dt, ncls):
    n, h, w = 2, 24, 40
    g = torch.Generator().manual_seed(6)
    code = L.dt_code(dt)
    a = rnd(dt, n, 64, h, w, gen=g)
    wt = torch.randn(ncls, 64, 1, 1, generator=g) / 8
    bias = torch.randn(ncls, generator=g)
    ar, wr, br = a.clone().requires_grad_(True), wt.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = F.conv2d(ar, wr, br)
    logits = torch.full((n, ncls, h, w), float("nan"), device=DEV)
    ad = to_nhwc(a, dt)
    L.call("fea_head_fwd", ad, 64, wt.to(DEV), bias.to(DEV), logits, n, h, w, 64, ncls, code, S())
    torch.cuda.synchronize()
    assert rel_l2(logits.cpu(), ref.detach()) < 1e-5        # fp32 math on both sides
    dl = torch.randn(n, ncls, h, w, generator=g)
    ref.backward(dl)
    nblk = L.call("fea_head_bwd_blocks", n, h, w)
    part = torch.empty(nblk * ncls * 65, device=DEV)
    da = torch.full((n, h, w, 64), float("nan"), dtype=dt, device=DEV)
    dW, db = torch.empty(ncls, 64, device=DEV), torch.empty(ncls, device=DEV)
    L.call("fea_head_bwd", dl.to(DEV), ad, 64, wt.to(DEV), da, 64, part, n, h, w, 64, ncls, code, S())
    L.call("fea_head_bwd_finalize", part, nblk, dW, db, 64, ncls, S())
    torch.cuda.synchronize()
    check(from_nhwc(da, 64), ar.grad, dt, "head bwd data")
    assert rel_l2(dW.cpu(), wr.grad.reshape(ncls, 64)) < 1e-5
    assert rel_l2(db.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 6, 24, 40), (1, 1, 16, 16), (3, 8, 32, 36), (2, 3, 6, 6)])
def test_fused_head_loss_matches_the_separate_kernels(dt, shape):
    """fea_head_loss_sums / fea_head_loss_bwd (logits and dlogits kept in registers) against the chain they replace:
    fea_head_fwd -> fea_bce_dice_sums -> fea_bce_dice_grad -> fea_head_bwd (+ finalize), and against the oracle's loss."""
    from oracle import unet_oracle as O
    n, c, h, w = shape
    code = L.dt_code(dt)
    g = torch.Generator().manual_seed(8)
    a = rnd(dt, n, 64, h, w, gen=g)
    wt = torch.randn(c, 64, generator=g) / 4
    bias = torch.randn(c, generator=g)
    tgt = (torch.rand(n, c, h, w, generator=g) < 0.2).to(torch.uint8)
    assert L.call("fea_head_loss_supported", h, w, c) == 1
    ad, wd, bd, td = to_nhwc(a, dt, ld=72, off=8), wt.to(DEV), bias.to(DEV), tgt.to(DEV)
    ap = view_ptr(ad, 8)
    # separate kernels
    logits = torch.empty(n, c, h, w, device=DEV)
    L.call("fea_head_fwd", ap, 72, wd, bd, logits, n, h, w, 64, c, code, S())
    part = torch.empty(L.call("fea_loss_blocks", n, c, h * w) * 4, device=DEV)
    sums = torch.empty(c, 4, dtype=torch.float64, device=DEV)
    loss = torch.empty(1, device=DEV)
    dl = torch.empty_like(logits)
    L.call("fea_bce_dice_sums", logits, td, sums, part, n, c, h * w, S())
    L.call("fea_bce_dice_grad", logits, td, sums, loss, dl, 1.0, n, c, h * w, S())
    nblk = L.call("fea_head_bwd_blocks", n, h, w)
    hpart = torch.empty(nblk * c * 65, device=DEV)
    da = torch.full((n, h, w, 64), float("nan"), dtype=dt, device=DEV)
    dW, db = torch.empty(c, 64, device=DEV), torch.empty(c, device=DEV)
    L.call("fea_head_bwd", dl, ap, 72, wd, da, 64, hpart, n, h, w, 64, c, code, S())
    L.call("fea_head_bwd_finalize", hpart, nblk, dW, db, 64, c, S())
    # fused
    lg2 = torch.full((n, c, h, w), float("nan"), device=DEV)
    sums2 = torch.empty(c, 4, dtype=torch.float64, device=DEV)
    part2 = torch.empty(nblk * c * 4, device=DEV)
    loss2 = torch.empty(1, device=DEV)
    hpart2 = torch.empty(nblk * c * 65, device=DEV)
    da2 = torch.full((n, h, w, 80), float("nan"), dtype=dt, device=DEV)
    dW2, db2 = torch.empty(c, 64, device=DEV), torch.empty(c, device=DEV)
    L.call("fea_head_loss_sums", ap, 72, wd, bd, td, lg2, sums2, part2, n, h, w, 64, c, code, S())
    L.call("fea_head_loss_bwd", ap, 72, wd, bd, td, sums2, loss2, view_ptr(da2, 16), 80, hpart2, 1.0, n, h, w, 64, c, code, S())
    L.call("fea_head_bwd_finalize", hpart2, nblk, dW2, db2, 64, c, S())
    torch.cuda.synchronize()
    mfma = dt == torch.bfloat16 and (h * w) % 16 == 0     # matrix-pipe form: fp32 weight as hi + lo bf16 (2^-16), other summation order
    if dt == torch.bfloat16 and not mfma:  # same kernel structure as fea_head_fwd's bf16 path: identical bits
        assert torch.equal(lg2, logits), "the fused kernel's logits differ from fea_head_fwd's"
    tol = 2e-5 if mfma else 1e-6
    assert rel_l2(lg2.cpu(), logits.cpu()) < tol, rel_l2(lg2.cpu(), logits.cpu())
    assert rel_l2(sums2.cpu(), sums.cpu()) < tol
    assert abs(loss2.item() - loss.item()) < tol * abs(loss.item())
    ref_loss = O.bce_dice_loss(logits.cpu(), tgt)
    assert abs(loss2.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    check(da2[..., 16:80], from_nhwc(da, 64).permute(0, 2, 3, 1), dt, "fused head backward: da")
    assert torch.isnan(da2[..., :16].float()).all()
    assert rel_l2(dW2.cpu(), dW.cpu()) < (1e-4 if mfma else 1e-5) and rel_l2(db2.cpu(), db.cpu()) < (1e-4 if mfma else 1e-5)


@pytest.mark.parametrize("shape", [(2, 6, 32, 48), (1, 1, 16, 16), (3, 6, 128, 96), (2, 3, 5, 7)])
def test_loss_mask_iou(shape):
    """BCE+Dice loss and its gradient within fp32 tolerance; mask and IoU
    counts BIT-EXACT (integer work) against torch boolean ops on identical logits."""
    from oracle import unet_oracle as O
    n, c, h, w = shape
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(n, c, h, w, generator=g) * 3
    logits[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-30, -1e-30])      # threshold edge cases
    target = (torch.rand(n, c, h, w, generator=g) < 0.1).to(torch.uint8)
    lr = logits.clone().requires_grad_(True)
    ref_loss = O.bce_dice_loss(lr, target)
    ref_loss.backward()
    ld, td = logits.to(DEV), target.to(DEV)
    nblk = L.call("fea_loss_blocks", n, c, h * w)
    part = torch.empty(nblk * 4, device=DEV)
    sums = torch.empty(c, 4, dtype=torch.float64, device=DEV)
    loss = torch.empty(1, device=DEV)
    dl = torch.empty_like(ld)
    L.call("fea_bce_dice_sums", ld, td, sums, part, n, c, h * w, S())
    L.call("fea_bce_dice_grad", ld, td, sums, loss, dl, 1.0, n, c, h * w, S())
    mask = torch.full((n, c, h, w), 7, dtype=torch.uint8, device=DEV)
    inter = torch.zeros(c, dtype=torch.int64, device=DEV)
    union = torch.zeros(c, dtype=torch.int64, device=DEV)
    L.call("fea_mask_iou", ld, td, mask, inter, union, n, c, h * w, S())
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * abs(ref_loss.item())
    assert rel_l2(dl.cpu(), lr.grad) < 1e-4
    pm, pi, pu = O.mask_iou_counts(logits, target)
    assert torch.equal(mask.cpu(), pm)
    assert torch.equal(inter.cpu(), pi) and torch.equal(union.cpu(), pu)
    grey = torch.randint(0, 256, (1000,), dtype=torch.uint8, generator=g)
    out = torch.empty(1000, dtype=torch.uint8, device=DEV)
    L.call("fea_binarise", grey.to(DEV), out, 1000, S())
    assert torch.equal(out.cpu(), O.binarise_mask(grey))


@pytest.mark.parametrize("dt", DTYPES)
def test_maxpool_backward_exact_ties_go_to_the_first_maximum(dt):
    """MaxPool2d(2) backward on windows with EXACT non-zero ties (models.py:24): the gradient goes to the first maximum
    in row-major window order, as torch's CPU kernel does.  Small integers are exact in bf16 and fp32;
    with scale 1, shift 0, mean 0, invstd 1 and zero coefficients, fea_bn_relu_bwd_apply reduces to ReLU-mask x pool
    routing, so the comparison is bit-exact."""
    code = L.dt_code(dt)
    n, c, h, w = 2, 64, 16, 24
    g = torch.Generator().manual_seed(23)
    y = torch.randint(-1, 3, (n, c, h, w), generator=g).float()           # {-1,0,1,2}: most windows tie
    y[0, :, 0:2, 0:2] = 2.0                                                # four-way tie
    y[0, :, 0:2, 2:4] = torch.tensor([[1.0, 2.0], [2.0, 0.0]])             # tie between positions 1 and 2
    y[0, :, 2:4, 0:2] = torch.tensor([[0.0, 1.0], [0.0, 1.0]])             # tie between positions 1 and 3
    dp = torch.randint(1, 8, (n, c, h // 2, w // 2), generator=g).float()
    yr = y.clone().requires_grad_(True)
    F.max_pool2d(torch.relu(yr), 2).backward(dp)
    ties = (F.max_pool2d(torch.relu(y), 2).repeat_interleave(2, 2).repeat_interleave(2, 3) == torch.relu(y)) & (y > 0)
    assert (ties.view(n, c, h // 2, 2, w // 2, 2).sum((3, 5)) > 1).float().mean() > 0.3, "test vector has too few ties"
    one, zero = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)
    coef = torch.zeros(c, 2, device=DEV)
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, c)
    part = torch.empty(nblk, c, 2, device=DEV)
    dy = torch.full((n, h, w, c), float("nan"), dtype=dt, device=DEV)
    L.call("fea_bn_relu_bwd_apply", None, 0, to_nhwc(dp, dt), to_nhwc(y, dt), one, zero, zero, one, coef, dy, part,
           n, h, w, c, code, S())
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(dy, c), yr.grad), "pool gradient not routed to the first maximum"
    # the reduce pass sees the same routing: sum g per channel
    L.call("fea_bn_relu_bwd_reduce", None, 0, to_nhwc(dp, dt), to_nhwc(y, dt), one, zero, zero, one, part, n, h, w, c, code, S())
    torch.cuda.synchronize()
    assert torch.equal(part[..., 0].double().sum(0).cpu(), yr.grad.double().sum((0, 2, 3)))


def test_bad_arguments_are_rejected():
    x = torch.zeros(64, device=DEV)
    with pytest.raises(L.FeaError):
        L.call("fea_conv3x3", x, 48, x, None, x, 64, None, 1, 8, 8, 48, 64, L.F32, S())       # Cin % 32
    with pytest.raises(L.FeaError):
        L.call("fea_conv3x3", None, 64, x, None, x, 64, None, 1, 8, 8, 64, 64, L.F32, S())     # null
    with pytest.raises(L.FeaError):
        L.call("fea_head_fwd", x, 64, x, x, x, 1, 8, 8, 64, 17, L.F32, S())                    # too many classes
    with pytest.raises(L.FeaError):
        L.call("fea_conv3x3_wgrad", x, 64, x, 64, x, x, 16, 1, 8, 8, 64, 64, L.F32, S())      # workspace too small


def test_large_partial_tables_are_folded_correctly():
    """>= 4096 rows: fea_bn_finalize / fea_reduce_partials_n fold the table in place first."""
    g = torch.Generator().manual_seed(21)
    rows, c = 5000, 64
    stats = torch.rand(rows, c, 2, generator=g)
    count = rows * 256
    s = stats.double().sum(0)
    mean = s[:, 0] / count
    var = (s[:, 1] / count - mean * mean).clamp(min=0)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    outs = [torch.empty(c, device=DEV) for _ in range(4)]
    L.call("fea_bn_finalize", stats.to(DEV), rows, count, gamma.to(DEV), beta.to(DEV), None, None, 0.1, 1e-5, *outs, c, S())
    torch.cuda.synchronize()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    assert rel_l2(outs[2], mean.float()) < 1e-6 and rel_l2(outs[3], invstd.float()) < 1e-6
    assert rel_l2(outs[0], (gamma.double() * invstd).float()) < 1e-6
    # first 32 channels of 128-channel-wide rows
    part = torch.rand(rows, 128, 2, generator=g)
    out = torch.empty(32, device=DEV)
    L.call("fea_reduce_partials_n", part.to(DEV), rows, 128, 32, 0, out, S())
    torch.cuda.synchronize()
    assert rel_l2(out, part[:, :32, 0].double().sum(0).float()) < 1e-6


@pytest.mark.parametrize("dt", DTYPES)
def test_pack_batch_equals_the_single_pack_calls(dt):
    """fea_pack_batch (one launch for all weight packs of a step) against fea_pack_conv3x3 / _frag / _upconv2x2 / _first
    called one by one on the same weights: bit-identical buffers."""
    import struct
    g = torch.Generator().manual_seed(11)
    code = L.BF16 if dt == torch.bfloat16 else L.F32
    jobs = [("fea_pack_conv3x3", 0, (128, 64, 0), torch.randn(128, 64, 3, 3, generator=g)),
            ("fea_pack_conv3x3", 0, (128, 64, 1), torch.randn(128, 64, 3, 3, generator=g)),
            ("fea_pack_upconv2x2", 2, (128, 64, 0), torch.randn(128, 64, 2, 2, generator=g)),
            ("fea_pack_upconv2x2", 2, (128, 64, 1), torch.randn(128, 64, 2, 2, generator=g))]
    if dt == torch.bfloat16:
        jobs += [("fea_pack_conv3x3_frag", 1, (64, 128, 0), torch.randn(64, 128, 3, 3, generator=g)),
                 ("fea_pack_conv3x3_frag", 1, (64, 64, 1), torch.randn(64, 64, 3, 3, generator=g)),
                 ("fea_pack_conv3x3_first", 3, (64, 3), torch.randn(64, 3, 3, 3, generator=g))]
    S = L.stream_ptr()
    recs, first, keep = [], 0, []
    for fn, kind, args, w in jobs:
        wd = w.to(DEV)
        numel = ((9 * args[1] + 31) // 32) * 64 * 32 if kind == 3 else w.numel()
        one = torch.zeros(numel, dtype=dt, device=DEV)
        many = torch.zeros(numel, dtype=dt, device=DEV)
        L.call(fn, wd, one, *args, code, S)
        a, b = args[0], args[1]
        c = args[2] if len(args) > 2 else 0
        recs.append(struct.pack("<QQiiiiii", wd.data_ptr(), many.data_ptr(), kind, a, b, c, first, 0))
        first += L.call("fea_pack_job_blocks", kind, a, b, c)
        keep.append((fn, wd, one, many))
    table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(DEV)
    L.call("fea_pack_batch", table, len(recs), first, code, S)
    torch.cuda.synchronize()
    for fn, _w, one, many in keep:
        assert torch.equal(one.view(torch.int16 if dt == torch.bfloat16 else torch.int32),
                           many.view(torch.int16 if dt == torch.bfloat16 else torch.int32)), fn


@pytest.mark.parametrize("cin", [3, 5])
@pytest.mark.parametrize("shape", [(2, 32, 64), (1, 40, 72), (3, 64, 96)])
def test_first_layer_wgrad_with_fused_bn_backward(cin, shape):
    """fea_conv3x3_first_wgrad_bnbwd == fea_bn_relu_bwd_apply followed by fea_conv3x3_first_wgrad (+ fea_reduce_partials for
    the conv bias gradient): the conv-output gradient is formed by the same fp32 expression and rounded to bf16 once, so
    the weight gradients agree to fp32 summation order; also against autograd on the CPU."""
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, h, w = shape
    c = 64
    g = torch.Generator().manual_seed(29)
    x = rnd(dt, n, cin, h, w, gen=g)
    y = (rnd(dt, n, c, h, w, gen=g) * 1.5 + 0.2).to(dt).float()
    da = rnd(dt, n, c, h, w, gen=g)
    gamma = torch.rand(c, generator=g) + 0.5
    gamma[::5] *= -1
    beta = torch.randn(c, generator=g) * 0.2
    mean = y.mean((0, 2, 3))
    invstd = torch.rsqrt(y.var((0, 2, 3), unbiased=False) + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    coef = torch.randn(c, 2, generator=g) * 1e-2
    dev = [t.to(DEV).contiguous() for t in (scale, shift, mean, invstd)]
    coefd = coef.to(DEV).contiguous()
    xd = x.to(DEV).contiguous()
    dad, yd = to_nhwc(da, dt, ld=c + 64, off=64), to_nhwc(y, dt)
    # separate kernels
    nblk = L.call("fea_bn_bwd_blocks", n, h, w, c)
    part = torch.full((max(nblk, 512), c, 2), float("nan"), device=DEV)
    dy = torch.full((n, h, w, c), float("nan"), dtype=dt, device=DEV)
    L.call("fea_bn_relu_bwd_apply", view_ptr(dad, 64), c + 64, None, yd, *dev, coefd, dy, part, n, h, w, c, code, S())
    db_ref = torch.empty(c, device=DEV)
    L.call("fea_reduce_partials", part, nblk, c, 0, db_ref, S())
    nbytes = L.load().fea_first_wgrad_ws_bytes(n, h, w, c)
    ws = torch.empty(nbytes // 4, device=DEV)
    dW_ref = torch.full((c, cin, 3, 3), float("nan"), device=DEV)
    L.call("fea_conv3x3_first_wgrad", dy, c, xd, dW_ref, ws, nbytes, n, cin, h, w, c, code, S())
    # fused
    rows = L.call("fea_first_wgrad_bnbwd_rows", n, h, w, c)
    assert 0 < rows <= 256
    part2 = torch.full((rows, c, 2), float("nan"), device=DEV)
    dW = torch.full((c, cin, 3, 3), float("nan"), device=DEV)
    L.call("fea_conv3x3_first_wgrad_bnbwd", view_ptr(dad, 64), c + 64, yd, c, *dev, coefd, xd, dW, part2, ws, nbytes,
           n, cin, h, w, c, code, S())
    db = torch.empty(c, device=DEV)
    L.call("fea_reduce_partials", part2, rows, c, 0, db, S())
    torch.cuda.synchronize()
    assert rel_l2(dW.cpu(), dW_ref.cpu()) < 1e-5, rel_l2(dW.cpu(), dW_ref.cpu())
    assert rel_l2(db.cpu(), db_ref.cpu()) < 1e-5
    # CPU reference from the stored bf16 gradient of the separate path
    wz = torch.zeros(c, cin, 3, 3, requires_grad=True)
    F.conv2d(x, wz, padding=1).backward(from_nhwc(dy, c))
    check(dW, wz.grad, dt, "fused first-layer wgrad", scale=0.5)


@pytest.mark.parametrize("shape", [(2, 6, 24, 40), (1, 1, 16, 16), (3, 4, 32, 32), (2, 5, 8, 12)])
def test_fused_head_loss_with_batchnorm_in_the_loader(shape):
    """fea_head_loss_sums_bn / fea_head_loss_bwd_bn (BatchNorm + ReLU of dec1.3 applied to the loaded pre-BN tensor) against
    fea_bn_relu_pool followed by fea_head_loss_sums / fea_head_loss_bwd: the activation is formed by the same expression and
    rounded to bf16 the same way, so everything is bit-identical."""
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, c, h, w = shape
    assert L.call("fea_head_loss_bn_supported", n, h, w, c, code) == 1
    assert L.call("fea_head_loss_bn_supported", n, h, w, 7, code) == 0 and L.call("fea_head_loss_bn_supported", n, h, w, c, L.F32) == 0
    g = torch.Generator().manual_seed(37)
    y = rnd(dt, n, 64, h, w, gen=g)
    scale = torch.rand(64, generator=g) + 0.5
    scale[::6] *= -1
    shift = torch.randn(64, generator=g) * 0.3
    wt = torch.randn(c, 64, generator=g) / 4
    bias = torch.randn(c, generator=g)
    tgt = (torch.rand(n, c, h, w, generator=g) < 0.2).to(torch.uint8)
    yd, wd, bd, td = to_nhwc(y, dt, ld=72, off=8), wt.to(DEV), bias.to(DEV), tgt.to(DEV)
    scd, shd = scale.to(DEV), shift.to(DEV)
    a = torch.full((n, h, w, 64), float("nan"), dtype=dt, device=DEV)
    # fea_bn_relu_pool takes a dense y
    ydense = to_nhwc(y, dt)
    L.call("fea_bn_relu_pool", ydense, scd, shd, a, 64, None, n, h, w, 64, code, S())
    nblk = L.call("fea_head_bwd_blocks", n, h, w)
    res = []
    for fused in (False, True):
        lg = torch.full((n, c, h, w), float("nan"), device=DEV)
        sums = torch.empty(c, 4, dtype=torch.float64, device=DEV)
        part = torch.empty(nblk * c * 4, device=DEV)
        loss = torch.empty(1, device=DEV)
        hpart = torch.empty(nblk * c * 65, device=DEV)
        da = torch.full((n, h, w, 64), float("nan"), dtype=dt, device=DEV)
        dW, db = torch.empty(c, 64, device=DEV), torch.empty(c, device=DEV)
        if fused:
            L.call("fea_head_loss_sums_bn", view_ptr(yd, 8), 72, scd, shd, wd, bd, td, lg, sums, part, n, h, w, 64, c, code, S())
            L.call("fea_head_loss_bwd_bn", view_ptr(yd, 8), 72, scd, shd, wd, bd, td, sums, loss, da, 64, hpart, 1.0, n, h, w, 64, c, code, S())
        else:
            L.call("fea_head_loss_sums", a, 64, wd, bd, td, lg, sums, part, n, h, w, 64, c, code, S())
            L.call("fea_head_loss_bwd", a, 64, wd, bd, td, sums, loss, da, 64, hpart, 1.0, n, h, w, 64, c, code, S())
        L.call("fea_head_bwd_finalize", hpart, nblk, dW, db, 64, c, S())
        torch.cuda.synchronize()
        res.append((lg, sums, loss, da, dW, db))
    for u, v in zip(*res):
        assert torch.equal(u, v)
    # and the activation itself against torch
    ref = torch.relu(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    check(from_nhwc(a, 64), ref, dt, "bn + relu")
    with pytest.raises(L.FeaError):
        L.call("fea_head_loss_sums_bn", view_ptr(yd, 8), 72, scd, shd, wd, bd, td, None, sums, part, n, h, w, 64, c, L.F32, S())


@pytest.mark.parametrize("shape", [(2, 6, 24, 40), (1, 1, 16, 16), (3, 4, 32, 32)])
def test_fused_head_loss_with_batchnorm_backward(shape):
    """fea_head_loss_bwd_bnred / fea_head_loss_bwd_bnapply (dec1.3's BatchNorm backward inside the head's backward, da never
    stored) against fea_head_loss_bwd_bn -> fea_bn_relu_bwd_reduce -> fea_bn_bwd_finalize -> fea_bn_relu_bwd_apply:
    dW / db / loss and the conv-output gradient dy bit-identical, the BatchNorm sums to fp32 summation order."""
    dt = torch.bfloat16
    code = L.dt_code(dt)
    n, c, h, w = shape
    g = torch.Generator().manual_seed(39)
    y = rnd(dt, n, 64, h, w, gen=g)
    gamma = torch.rand(64, generator=g) + 0.5
    gamma[::6] *= -1
    beta = torch.randn(64, generator=g) * 0.3
    mean = y.mean((0, 2, 3))
    invstd = torch.rsqrt(y.var((0, 2, 3), unbiased=False) + 1e-5)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    wt = torch.randn(c, 64, generator=g) / 4
    bias = torch.randn(c, generator=g)
    tgt = (torch.rand(n, c, h, w, generator=g) < 0.2).to(torch.uint8)
    yd, wd, bd, td = to_nhwc(y, dt), wt.to(DEV), bias.to(DEV), tgt.to(DEV)
    cf = [t.to(DEV).contiguous() for t in (scale, shift, mean, invstd)]
    gam = gamma.to(DEV)
    nblk = L.call("fea_head_bwd_blocks", n, h, w)
    sums = torch.empty(c, 4, dtype=torch.float64, device=DEV)
    part = torch.empty(nblk * c * 4, device=DEV)
    L.call("fea_head_loss_sums_bn", yd, 64, cf[0], cf[1], wd, bd, td, None, sums, part, n, h, w, 64, c, code, S())
    # reference chain
    loss_r = torch.empty(1, device=DEV)
    hpart = torch.empty(nblk * c * 65, device=DEV)
    da = torch.full((n, h, w, 64), float("nan"), dtype=dt, device=DEV)
    L.call("fea_head_loss_bwd_bn", yd, 64, cf[0], cf[1], wd, bd, td, sums, loss_r, da, 64, hpart, 1.0, n, h, w, 64, c, code, S())
    dW_r, db_r = torch.empty(c, 64, device=DEV), torch.empty(c, device=DEV)
    L.call("fea_head_bwd_finalize", hpart, nblk, dW_r, db_r, 64, c, S())
    nb = L.call("fea_bn_bwd_blocks", n, h, w, 64)
    bpart = torch.zeros(max(nb, nblk), 64, 2, device=DEV)
    L.call("fea_bn_relu_bwd_reduce", da, 64, None, yd, *cf, bpart, n, h, w, 64, code, S())
    red_ref = bpart[:nb].double().sum(0).cpu()
    dgam_r, dbet_r, coef_r = torch.empty(64, device=DEV), torch.empty(64, device=DEV), torch.empty(64, 2, device=DEV)
    L.call("fea_bn_bwd_finalize", bpart, nb, n * h * w, gam, cf[3], dgam_r, dbet_r, coef_r, 64, S())
    dy_r = torch.full((n, h, w, 64), float("nan"), dtype=dt, device=DEV)
    L.call("fea_bn_relu_bwd_apply", da, 64, None, yd, *cf, coef_r, dy_r, bpart, n, h, w, 64, code, S())
    dbias_r = torch.empty(64, device=DEV)
    L.call("fea_reduce_partials", bpart, nb, 64, 0, dbias_r, S())
    # fused chain
    loss = torch.empty(1, device=DEV)
    hpart2 = torch.empty(nblk * c * 65, device=DEV)
    bpart2 = torch.full((nblk, 64, 2), float("nan"), device=DEV)
    L.call("fea_head_loss_bwd_bnred", yd, 64, *cf, wd, bd, td, sums, loss, hpart2, bpart2, 1.0, n, h, w, 64, c, code, S())
    dW, db = torch.empty(c, 64, device=DEV), torch.empty(c, device=DEV)
    L.call("fea_head_bwd_finalize", hpart2, nblk, dW, db, 64, c, S())
    torch.cuda.synchronize()
    red = bpart2.double().sum(0).cpu()
    dgam, dbet, coef = torch.empty(64, device=DEV), torch.empty(64, device=DEV), torch.empty(64, 2, device=DEV)
    L.call("fea_bn_bwd_finalize", bpart2, nblk, n * h * w, gam, cf[3], dgam, dbet, coef, 64, S())
    # the apply pass is compared with the reference's coefficients, so that dy can be compared bit for bit
    dy = torch.full((n, h, w, 80), float("nan"), dtype=dt, device=DEV)
    L.call("fea_head_loss_bwd_bnapply", yd, 64, *cf, coef_r, wd, bd, td, sums, view_ptr(dy, 16), 80, bpart2, 1.0, n, h, w, 64, c, code, S())
    dbias = torch.empty(64, device=DEV)
    L.call("fea_reduce_partials", bpart2, nblk, 64, 0, dbias, S())
    torch.cuda.synchronize()
    assert torch.equal(loss, loss_r) and torch.equal(dW, dW_r) and torch.equal(db, db_r)
    assert rel_l2(red[:, 0], red_ref[:, 0]) < 1e-5 and rel_l2(red[:, 1], red_ref[:, 1]) < 1e-4, (rel_l2(red[:, 0], red_ref[:, 0]), rel_l2(red[:, 1], red_ref[:, 1]))
    assert rel_l2(dgam.cpu(), dgam_r.cpu()) < 1e-4 and rel_l2(dbet.cpu(), dbet_r.cpu()) < 1e-5
    assert torch.equal(dy[..., 16:].contiguous().view(torch.int16), dy_r.view(torch.int16)), "conv-output gradient differs from fea_bn_relu_bwd_apply's"
    assert torch.isnan(dy[..., :16].float()).all()
    assert (dbias - dbias_r).abs().max().item() <= 1e-4 * dy_r.float().abs().sum((0, 1, 2)).max().item() + 1e-6
