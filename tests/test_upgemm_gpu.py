"""Persistent GEMM of the transposed-convolution pair (fea_amd/csrc/upgemm.hip) behind fea_upconv2x2_fwd / fea_upconv2x2_bwd_data*.

nn.ConvTranspose2d(ci, co, 2, stride=2) (models.py:15-21) against F.conv_transpose2d / autograd on the CPU, bf16-rounded inputs, at the
bounds of tests/test_kernels_gpu.py.  The kernel walks tiles through a four-stage LDS ring that runs on across tiles, so the cases
force several tiles per workgroup (FEA_AMD_UPG_WALKERS limits the workgroups), several groups of four k-steps per tile, every column
group count, and the image-width branches of the gather (W <= 128, W = 256); one case is in the bench regime (default launch, four
tiles per workgroup).  Each case also runs the implicit-GEMM path (FEA_AMD_UPGEMM=0) on the same inputs and compares the two.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from fea_amd import _lib as L
from tests.test_kernels_gpu import DEV, S, check, from_nhwc, rnd, to_nhwc

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# N, H, W, Cin, Cout, walkers (None = default launch)
FWD = [(2, 32, 32, 128, 64, None),        # 8 tiles, one column group, one tile per workgroup
       (2, 32, 32, 128, 64, 8),           # the same on 8 workgroups... 1 tile each; see below for more
       (4, 32, 32, 256, 128, 8),          # 16 tiles on 8 walkers x 2 column groups: 2 tiles per workgroup, 2 groups of k-steps
       (4, 16, 16, 128, 256, 8),          # W = 16, four column groups
       (1, 64, 256, 128, 64, 8),          # W = 256: 64 tiles, 8 per workgroup, tile = one image row
       (2, 16, 64, 512, 128, 8),          # K = 512: four groups of k-steps
       (64, 64, 64, 128, 64, None)]       # bench regime: 1,024 tiles on 256 workgroups


@pytest.mark.parametrize("case", FWD)
def test_upconv_fwd_persistent_gemm(case):
    n, h, w, ci, co, walkers = case
    g = torch.Generator().manual_seed(7)
    code = L.dt_code(BF)
    x = rnd(BF, n, ci, h, w, gen=g)
    wt = (torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5).to(BF).float()
    bias = torch.randn(co, generator=g)
    ref = F.conv_transpose2d(x, wt, bias, stride=2)
    wp = torch.empty(4 * ci * co, dtype=BF, device=DEV)
    L.call("fea_pack_upconv2x2", wt.to(DEV), wp, ci, co, 0, code, S())
    xd = to_nhwc(x, BF)
    outs = []
    for on in (1, 0):
        cat = torch.full((n, 2 * h, 2 * w, 2 * co), float("nan"), dtype=BF, device=DEV)
        with _env(FEA_AMD_UPGEMM=on, FEA_AMD_UPG_WALKERS=walkers):
            L.call("fea_upconv2x2_fwd", xd, ci, wp, bias.to(DEV), cat, 2 * co, n, h, w, ci, co, code, S())
        torch.cuda.synchronize()
        assert torch.isnan(cat[..., co:].float()).all(), "skip half of the concat buffer was touched"
        outs.append(cat[..., :co].clone())
        check(from_nhwc(cat, co, 0), ref, BF, f"upconv fwd (persistent GEMM = {on})")
    # both paths accumulate the same products in fp32 (different order) and round once to bf16
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    assert d <= 2 ** -6 * max(1.0, ref.abs().max().item()), d


def test_upconv_fwd_persistent_gemm_slice_input_no_bias_and_split():
    """x as a channel slice of a wider buffer (ldx > Cin), no bias, and the persistent work split of the data-parallel trainer
    (fea_set_persistent_split(4): four times the walkers, a quarter of the tile range each)."""
    n, h, w, ci, co = 4, 32, 64, 128, 64
    g = torch.Generator().manual_seed(8)
    code = L.dt_code(BF)
    x = rnd(BF, n, ci, h, w, gen=g)
    wt = (torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5).to(BF).float()
    ref = F.conv_transpose2d(x, wt, None, stride=2)
    wp = torch.empty(4 * ci * co, dtype=BF, device=DEV)
    L.call("fea_pack_upconv2x2", wt.to(DEV), wp, ci, co, 0, code, S())
    wide = torch.full((n, h, w, ci + 64), float("nan"), dtype=BF, device=DEV)
    wide[..., 64:] = to_nhwc(x, BF)
    xs = wide.data_ptr() + 64 * 2
    lib = L.load()
    old = lib.fea_get_persistent_split()
    try:
        for split in (1, 4):
            assert lib.fea_set_persistent_split(split) == 0
            y = torch.full((n, 2 * h, 2 * w, co), float("nan"), dtype=BF, device=DEV)
            with _env(FEA_AMD_UPGEMM=1, FEA_AMD_UPG_WALKERS=None):
                L.call("fea_upconv2x2_fwd", xs, ci + 64, wp, None, y, co, n, h, w, ci, co, code, S())
            torch.cuda.synchronize()
            check(from_nhwc(y, co), ref, BF, f"upconv fwd, sliced input, split {split}")
    finally:
        lib.fea_set_persistent_split(old)
