"""Parity in the regime bench.py runs the kernels in.

At batch 64 x 512^2 every persistent / multi-tile kernel walks hundreds of tiles per workgroup: steady-state ring
parity flips, "the ring never drains / read the next tile ahead", one LDS-DMA piece per k-step.  The kernel tests of
tests/test_kernels_gpu.py top out at <= 2 tiles per workgroup, so a bug that only shows in the steady state (and is
deterministic and linear, hence invisible to tests/test_fullsize_gpu.py) would pass there.  Here:

  * kernel level: shapes where each workgroup walks >= 4 tiles, against F.conv2d / its weight gradient on the CPU
    (nn.Conv2d semantics, models.py:28,31) on bf16-rounded inputs, at the same bounds as the small shapes
    (4e-3 L2 / 2e-2 peak; weight gradients at half of that);
  * model level: BASELINE configs[2] at FULL size (batch 64, 512 x 512): the bf16 engine against the fp32 engine
    (which is pinned to the oracle and the reference goldens at 1e-4, tests/test_unet_gpu.py) on the same seed and
    batch -- logits, loss, every parameter gradient, BatchNorm running statistics -- with stated bounds, and a
    per-layer error table written for profiles/r3_bf16_vs_fp32_fullsize.txt.
"""
import os

import pytest
import torch
import torch.nn.functional as F

import fea_amd
from fea_amd import _lib as L
from fea_amd import functional as Fn
from fea_amd.data import synthetic_batch
from tests.helpers import max_rel, rel_l2
from tests.test_kernels_gpu import DEV, S, check, from_nhwc, rnd, to_nhwc, view_ptr

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _tiles_per_workgroup_p(n, h, w, co):
    """Tiles each persistent workgroup walks: 8x32-pixel tiles, 256 workgroups shared by the 64-column groups."""
    return n * (h // 8) * (w // 32) / (256 // max(1, co // 64))


# N, H, W, Cin, Cout: 1,024 pixel tiles -> 4 (64 columns) / 8 (128 columns) tiles per persistent workgroup
SHAPES_P_DEEP = [(16, 128, 128, 64, 64), (16, 128, 128, 128, 128), (16, 128, 128, 128, 64)]


@pytest.mark.parametrize("shape", SHAPES_P_DEEP)
def test_persistent_conv_many_tiles_per_workgroup(shape):
    """fea_conv3x3p (EPI 0 + BN sums), its plain data-gradient form, fea_conv3x3p_dgrad_bnred (EPI 2) and
    fea_conv3x3p_bias_relu (EPI 3, with pool) where every workgroup is in the steady state of its rings."""
    n, h, w, ci, co = shape
    code = L.dt_code(BF)
    rows = L.call("fea_conv3x3p_rows", n, h, w, ci, co, code)
    assert rows > 0 and _tiles_per_workgroup_p(n, h, w, co) >= 4
    g = torch.Generator().manual_seed(41)
    x = rnd(BF, n, ci, h, w, gen=g)
    wt = (rnd(BF, co, ci, 3, 3, gen=g) / (3 * ci ** 0.5)).to(BF).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    xd = to_nhwc(x, BF)
    wp = torch.empty(9 * co * ci, dtype=BF, device=DEV)
    L.call("fea_pack_conv3x3_frag", wt.to(DEV), wp, co, ci, 0, code, S())
    # EPI 0
    y = torch.full((n, h, w, co), float("nan"), dtype=BF, device=DEV)
    stats = torch.full((rows, co, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3p", xd, ci, wp, bias.to(DEV), y, co, stats, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(y, co), ref, BF, "persistent conv fwd, steady state")
    s = stats.double().sum(0).cpu()
    ref_s = torch.stack([ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))], 1)
    assert rel_l2(s, ref_s) < 1e-5, "BN partial sums"
    # EPI 3: bias + ReLU + pool
    a = torch.full((n, h, w, co), float("nan"), dtype=BF, device=DEV)
    pool = torch.full((n, h // 2, w // 2, co), float("nan"), dtype=BF, device=DEV)
    L.call("fea_conv3x3p_bias_relu", xd, ci, wp, bias.to(DEV), a, co, pool, co, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    got = from_nhwc(a, co)
    check(got, F.relu(ref), BF, "persistent conv + bias + relu, steady state")
    assert torch.equal(from_nhwc(pool, co), F.max_pool2d(got, 2))
    del ref, a, pool, y
    # data gradient (plain) and EPI 2 (fused BatchNorm-backward sums), where the transposed shape is served
    rows_d = L.call("fea_conv3x3p_rows", n, h, w, co, ci, code)
    if rows_d == 0:
        return
    dy = rnd(BF, n, co, h, w, gen=g)
    ref_dx = F.conv_transpose2d(dy, wt, padding=1)
    wpd = torch.empty(9 * co * ci, dtype=BF, device=DEV)
    L.call("fea_pack_conv3x3_frag", wt.to(DEV), wpd, co, ci, 1, code, S())
    dyd = to_nhwc(dy, BF)
    dx = torch.full((n, h, w, ci), float("nan"), dtype=BF, device=DEV)
    L.call("fea_conv3x3p", dyd, co, wpd, None, dx, ci, None, n, h, w, co, ci, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(dx, ci), ref_dx, BF, "persistent conv dgrad, steady state")
    yb = rnd(BF, n, ci, h, w, gen=g)
    scale, shift = torch.randn(ci, generator=g), torch.randn(ci, generator=g)
    mean, invstd = torch.randn(ci, generator=g), torch.rand(ci, generator=g) + 0.5
    # the sums the kernel must produce, from the CPU reference gradient (not from another HIP kernel):
    # g = da * [relu active], s0 = sum g, s1 = sum g * xhat
    act = (yb * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    da_q = from_nhwc(dx, ci)                                                  # the bf16 gradient the kernel itself stored
    gq = (da_q * act).double()
    xhat = ((yb - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)).double()
    want = torch.stack([gq.sum((0, 2, 3)), (gq * xhat).sum((0, 2, 3))], 1)
    coef = [t.to(DEV) for t in (scale, shift, mean, invstd)]
    da = torch.full((n, h, w, ci), float("nan"), dtype=BF, device=DEV)
    part = torch.full((rows_d, ci, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3p_dgrad_bnred", dyd, co, wpd, da, ci, to_nhwc(yb, BF), ci, *coef, part, n, h, w, co, ci, code, S())
    torch.cuda.synchronize()
    assert torch.equal(da.view(torch.int16), dx.view(torch.int16)), "fused kernel changed the data gradient"
    got = part.double().sum(0).cpu()
    # the kernel sums the gradient it stored (bf16), as the reference sums here: fp32 accumulation order is the only difference
    assert rel_l2(got[:, 0], want[:, 0]) < 1e-5 and rel_l2(got[:, 1], want[:, 1]) < 1e-4, (rel_l2(got[:, 0], want[:, 0]), rel_l2(got[:, 1], want[:, 1]))


@pytest.mark.parametrize("shape,tiles_per_split,splits", [((16, 128, 128, 64, 64), 4, 256),     # one 64x64 block, 256 pixel splits
                                                          ((4, 32, 32, 512, 512), 4, 4),       # 64 blocks x 4 splits x 4 tiles, deep channels
                                                          ((16, 64, 128, 128, 64), 4, 128)])   # 2 blocks
def test_conv3x3_wgrad_many_tiles_per_split(shape, tiles_per_split, splits):
    """wgrad2_kernel<5,true> with several tiles per split: both LDS stages in use, the next tile's LDS-DMA pieces issued
    behind the MFMAs of the current one, slab reduction over the splits.  Against autograd's weight gradient."""
    n, h, w, ci, co = shape
    ntiles = n * (h // 8) * (w // 32)
    want = max(1, 256 // ((co // 64) * (ci // 64)))
    assert (ntiles + want - 1) // want == tiles_per_split and (ntiles + tiles_per_split - 1) // tiles_per_split == splits
    g = torch.Generator().manual_seed(42)
    x = rnd(BF, n, ci, h, w, gen=g)
    dy = rnd(BF, n, co, h, w, gen=g)
    wz = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x, wz, padding=1).backward(dy)
    code = L.dt_code(BF)
    nbytes = L.load().fea_wgrad_ws_bytes(0, n, h, w, co, ci, code)
    ws = torch.empty(max(nbytes, 16) // 4, device=DEV)
    dW = torch.full((co, ci, 3, 3), float("nan"), device=DEV)
    L.call("fea_conv3x3_wgrad", to_nhwc(dy, BF), co, to_nhwc(x, BF), ci, dW, ws, nbytes, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    check(dW, wz.grad, BF, "conv3x3 wgrad, several tiles per split", scale=0.5)
    # twice the same launch: the slab reduction has a fixed order
    dW2 = torch.empty_like(dW)
    L.call("fea_conv3x3_wgrad", to_nhwc(dy, BF), co, to_nhwc(x, BF), ci, dW2, ws, nbytes, n, h, w, ci, co, code, S())
    torch.cuda.synchronize()
    assert torch.equal(dW, dW2)


@pytest.mark.parametrize("shape", [(2, 32, 64, 512, 512), (4, 64, 64, 256, 256)])
def test_deep_dgrad_bnred_against_cpu_reference(shape):
    """fea_conv3x3_dgrad_bnred (igemm EPI 2) on the deep shapes of the step (512 -> 512, 256 -> 256) against
    F.conv_transpose2d and the BatchNorm-backward sums computed on the CPU from the kernel's own stored gradient."""
    n, h, w, ck, cn = shape
    code = L.dt_code(BF)
    g = torch.Generator().manual_seed(43)
    dy = rnd(BF, n, ck, h, w, gen=g)
    wt = (rnd(BF, ck, cn, 3, 3, gen=g) / (3 * ck ** 0.5)).to(BF).float()
    y = rnd(BF, n, cn, h, w, gen=g)
    scale, shift = torch.randn(cn, generator=g), torch.randn(cn, generator=g)
    mean, invstd = torch.randn(cn, generator=g), torch.rand(cn, generator=g) + 0.5
    ref_da = F.conv_transpose2d(dy, wt, padding=1)
    wpd = torch.empty(9 * ck * cn, dtype=BF, device=DEV)
    L.call("fea_pack_conv3x3", wt.to(DEV), wpd, ck, cn, 1, code, S())
    coef = [t.to(DEV) for t in (scale, shift, mean, invstd)]
    ntiles = L.call("fea_conv_num_tiles", n, h, w)
    da = torch.full((n, h, w, cn), float("nan"), dtype=BF, device=DEV)
    part = torch.full((ntiles, cn, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3_dgrad_bnred", to_nhwc(dy, BF), ck, wpd, da, cn, to_nhwc(y, BF), cn, *coef, part, n, h, w, ck, cn, code, S())
    torch.cuda.synchronize()
    da_q = from_nhwc(da, cn)
    check(da_q, ref_da, BF, "deep dgrad (EPI 2)")
    act = (y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
    gq = (da_q * act).double()
    xhat = ((y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)).double()
    want = torch.stack([gq.sum((0, 2, 3)), (gq * xhat).sum((0, 2, 3))], 1)
    got = part.double().sum(0).cpu()
    assert rel_l2(got[:, 0], want[:, 0]) < 1e-5 and rel_l2(got[:, 1], want[:, 1]) < 1e-4, (rel_l2(got[:, 0], want[:, 0]), rel_l2(got[:, 1], want[:, 1]))


def test_bn_backward_sums_with_a_large_mean_to_std_ratio():
    """the fused BatchNorm-backward sums are formed as invstd * (sum g*y - mean * sum g); with |mean| >> std
    of the pre-BN tensor that subtraction cancels.  y ~ N(10, 1): dgamma's sum must still hold 1e-4 against fp64."""
    n, h, w, c = 4, 64, 64, 64
    code = L.dt_code(BF)
    g = torch.Generator().manual_seed(44)
    dy = rnd(BF, n, c, h, w, gen=g)
    wt = (rnd(BF, c, c, 3, 3, gen=g) / (3 * c ** 0.5)).to(BF).float()
    y = (torch.randn(n, c, h, w, generator=g) + 10.0).to(BF).float()
    mean = y.mean((0, 2, 3))
    invstd = torch.rsqrt(y.var((0, 2, 3), unbiased=False) + 1e-5)
    gamma = torch.rand(c, generator=g) + 0.5
    scale, shift = gamma * invstd, 0.1 - mean * gamma * invstd            # bn(y) = gamma * xhat + 0.1
    coef = [t.to(DEV) for t in (scale, shift, mean, invstd)]
    yd, dyd = to_nhwc(y, BF), to_nhwc(dy, BF)
    for frag, fn, rows in ((1, "fea_conv3x3p_dgrad_bnred", L.call("fea_conv3x3p_rows", n, h, w, c, c, code)),
                           (0, "fea_conv3x3_dgrad_bnred", L.call("fea_conv_num_tiles", n, h, w))):
        wpd = torch.empty(9 * c * c, dtype=BF, device=DEV)
        L.call("fea_pack_conv3x3_frag" if frag else "fea_pack_conv3x3", wt.to(DEV), wpd, c, c, 1, code, S())
        da = torch.empty((n, h, w, c), dtype=BF, device=DEV)
        part = torch.full((rows, c, 2), float("nan"), device=DEV)
        L.call(fn, dyd, c, wpd, da, c, yd, c, *coef, part, n, h, w, c, c, code, S())
        torch.cuda.synchronize()
        # reference sums in fp64 from the gradient the kernel stored (bf16), which is what it sums
        check(from_nhwc(da, c), F.conv_transpose2d(dy, wt, padding=1), BF, fn)
        act = (y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)) > 0
        gq = (from_nhwc(da, c) * act).double()
        xhat = ((y - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)).double()
        want = torch.stack([gq.sum((0, 2, 3)), (gq * xhat).sum((0, 2, 3))], 1)
        got = part.double().sum(0).cpu()
        e0, e1 = rel_l2(got[:, 0], want[:, 0]), rel_l2(got[:, 1], want[:, 1])
        assert e0 < 1e-4 and e1 < 1e-4, (fn, e0, e1)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[2] at full size: bf16 engine against the (oracle-pinned) fp32 engine

FULL_N, FULL_S = 64, 512
# Stated bounds of the bf16 path at batch 64 x 512^2 against fp32 on the same seed and batch (kinks NOT pinned: the two
# runs take their own ReLU / max-pool decisions).  Observed values are in profiles/r3_bf16_vs_fp32_fullsize.txt.
BOUND_LOGITS_L2 = 2e-2
BOUND_LOSS_REL = 5e-3
BOUND_GRAD_WORST = 2.5e-1        # bottleneck / enc4: nine to ten layers of un-pinned kinks above them (observed 1.8e-1)
BOUND_GRAD_MEDIAN = 3e-2         # observed 1.4e-2
BOUND_GRAD_SHALLOW = 3e-2        # enc1, enc2, dec1, dec2 (the persistent-kernel levels): observed <= 1.6e-2
BOUND_GRAD_PROJECTION = 5e-2     # |<bf16, fp32> / <fp32, fp32> - 1|: the difference is noise around the fp32 gradient, not a scale error
BOUND_BN_STATS = 1e-2


def _cancelled(k):
    p = k.split(".")
    return len(p) == 3 and p[1] in ("0", "3") and p[2] == "bias"


def _one_step(dtype, x, t):
    torch.manual_seed(0)
    m = fea_amd.FloorPlanSegmentationModel(compute_dtype=dtype).cuda().train()
    eng = m._engine()
    logits = eng.forward(x, True)
    loss, dl = Fn.bce_dice_loss_and_grad(logits, t)
    eng.backward(dl)
    torch.cuda.synchronize()
    out = {"logits": logits.clone(), "loss": float(loss),
           "grads": {k: eng.grad_views[k].clone() for k in eng.grad_views},
           "bufs": {k: v.clone() for k, v in m.named_buffers() if not k.endswith("num_batches_tracked")}}
    del m, eng, logits, dl
    torch.cuda.empty_cache()
    return out


def test_config3_bf16_against_fp32_engine_at_full_size():
    x, t = synthetic_batch(FULL_N, FULL_S, FULL_S, seed=99)
    x = (x + 0.05 * torch.randn(x.shape, generator=torch.Generator().manual_seed(2))).cuda()
    t = t.cuda()
    ref = _one_step("fp32", x, t)
    got = _one_step("bf16", x, t)
    lines = [f"# bf16 engine vs fp32 engine, one training step (forward, BCE+Dice, backward), batch {FULL_N} x {FULL_S}x{FULL_S}, same seed and batch",
             "# rel_l2 = ||bf16 - fp32|| / ||fp32||; kinks (ReLU masks, pool arg-max) NOT pinned"]
    e_log = rel_l2(got["logits"], ref["logits"])
    e_loss = abs(got["loss"] - ref["loss"]) / abs(ref["loss"])
    agree = ((got["logits"] > 0) == (ref["logits"] > 0)).float().mean().item()
    lines += [f"logits rel_l2 {e_log:.3e}   peak-rel {max_rel(got['logits'], ref['logits']):.3e}   masks equal on {agree:.5f} of the pixels",
              f"loss   bf16 {got['loss']:.6f}  fp32 {ref['loss']:.6f}  rel {e_loss:.3e}", "# parameter gradients"]
    errs, viol = [], []
    for k, gr in ref["grads"].items():
        if _cancelled(k):
            continue
        if k.startswith("upconv") and k.endswith(".bias"):
            # nearly cancelled by the following BatchNorm (tests/test_unet_gpu.py): on the scale of the layer's weight gradient
            e = float((got["grads"][k] - gr).norm()) / float(ref["grads"][k.replace(".bias", ".weight")].norm())
            lines.append(f"  {k:24s} {e:.3e}  (relative to the layer's weight gradient)")
            if e >= 1e-2:
                viol.append((k, 'upconv bias', e))
            continue
        e = rel_l2(got["grads"][k], gr)
        proj = float((got["grads"][k].double() * gr.double()).sum() / (gr.double() ** 2).sum())
        errs.append((e, k))
        lines.append(f"  {k:24s} {e:.3e}   projection {proj:.4f}")
        if abs(proj - 1.0) >= BOUND_GRAD_PROJECTION:
            viol.append((k, 'projection', proj))
        if k.split(".")[0] in ("enc1", "enc2", "dec1", "dec2"):
            if e >= BOUND_GRAD_SHALLOW:
                viol.append((k, 'levels 0-1', e))
    errs.sort()
    lines.append(f"# worst {errs[-1][1]} {errs[-1][0]:.3e}   median {errs[len(errs) // 2][0]:.3e}")
    lines.append("# BatchNorm running statistics after the step")
    e_bn = []
    for k, b in ref["bufs"].items():
        e = rel_l2(got["bufs"][k], b)
        e_bn.append((e, k))
        lines.append(f"  {k:32s} {e:.3e}")
    lines.append(f"# bounds asserted: logits {BOUND_LOGITS_L2:g}, loss {BOUND_LOSS_REL:g}, gradients worst {BOUND_GRAD_WORST:g} / median "
                 f"{BOUND_GRAD_MEDIAN:g} / levels 0-1 {BOUND_GRAD_SHALLOW:g} / projection {BOUND_GRAD_PROJECTION:g}, BatchNorm statistics {BOUND_BN_STATS:g}")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "r3_bf16_vs_fp32_fullsize.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines))
    assert not viol, viol
    assert e_log < BOUND_LOGITS_L2, e_log
    assert e_loss < BOUND_LOSS_REL, e_loss
    assert errs[-1][0] < BOUND_GRAD_WORST and errs[len(errs) // 2][0] < BOUND_GRAD_MEDIAN, (errs[-1], errs[len(errs) // 2])
    assert max(e_bn)[0] < BOUND_BN_STATS, max(e_bn)
    # masks: flips only where the fp32 logit is inside the bf16 error band
    flips = (got["logits"] > 0) != (ref["logits"] > 0)
    if flips.any():
        assert float(ref["logits"][flips].abs().max()) < 0.25
    assert agree >= 0.98


@pytest.mark.parametrize("cin,n,hw", [(3, 2, (512, 512)), (5, 2, (512, 384))])
def test_first_layer_persistent_many_tiles_per_workgroup(cin, n, hw):
    """fea_conv3x3_first on its persistent kernel (first_conv.hip) where a workgroup walks several tiles: the halo of the next
    tile prefetched into registers, two LDS halo buffers, BatchNorm sums kept in registers across tiles and across image
    borders.  Against F.conv2d on the CPU (Conv2d(in_channels, 64, 3, padding=1), models.py:28), and against the
    one-tile-per-workgroup kernel (FEA_AMD_FIRST_P=0) on the same input."""
    h, w = hw
    co, code = 64, L.dt_code(BF)
    rows = L.call("fea_conv3x3_first_rows", n, cin, h, w, co, code)
    ntiles8 = n * (h // 8) * (w // 32)
    # (the regime this test is about needs the library's default split; under FEA_AMD_PERSISTENT_SPLIT=4, the
    # data-parallel setting, the same shapes give one tile per walker and the comparison below still has to hold)
    assert rows > 0 and (L.load().fea_get_persistent_split() > 1 or ntiles8 / (rows / 4) >= 2), (rows, ntiles8)
    g = torch.Generator().manual_seed(43)
    x = rnd(BF, n, cin, h, w, gen=g)
    wt = (rnd(BF, co, cin, 3, 3, gen=g) / (3 * cin ** 0.5)).to(BF).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv2d(x, wt, bias, padding=1)
    wp = torch.empty((9 * cin + 31) // 32 * co * 32, dtype=BF, device=DEV)
    L.call("fea_pack_conv3x3_first", wt.to(DEV), wp, co, cin, code, S())
    xd = x.to(DEV).contiguous()
    y = torch.full((n, h, w, co), float("nan"), dtype=BF, device=DEV)
    stats = torch.full((rows, co, 2), float("nan"), device=DEV)
    L.call("fea_conv3x3_first", xd, wp, bias.to(DEV), y, co, stats, n, cin, h, w, co, code, S())
    torch.cuda.synchronize()
    check(from_nhwc(y, co), ref, BF, "persistent first-layer conv")
    s = stats.double().sum(0).cpu()
    ref_s = torch.stack([ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))], 1)
    assert rel_l2(s, ref_s) < 1e-5, "BN partial sums"
    os.environ["FEA_AMD_FIRST_P"] = "0"
    try:
        assert L.call("fea_conv3x3_first_rows", n, cin, h, w, co, code) == 0
        y0 = torch.full((n, h, w, co), float("nan"), dtype=BF, device=DEV)
        st0 = torch.full((L.call("fea_conv_num_tiles", n, h, w), co, 2), float("nan"), device=DEV)
        L.call("fea_conv3x3_first", xd, wp, bias.to(DEV), y0, co, st0, n, cin, h, w, co, code, S())
        torch.cuda.synchronize()
    finally:
        del os.environ["FEA_AMD_FIRST_P"]
    # the same products in the same k order per output element, bias added at the start instead of the end: equal up to fp32 rounding
    assert (y.float() - y0.float()).abs().max().item() <= 2 ** -6 * max(1.0, ref.abs().max().item())
    assert rel_l2(s, st0.double().sum(0).cpu()) < 1e-6
