"""Input-pipeline oracle (oracle/preprocess_oracle.py) against the golden vectors made with
Pillow itself (oracle/make_golden_preprocess.py), and the library's HOST-side helpers
(coefficient tables, normalisation LUT: no GPU work) against the oracle.  Bit-exact."""
import os

import numpy as np
import pytest
import torch

from fea_amd import _lib as L
from oracle import preprocess_oracle as P

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_golden.npz"))
NAMES = [str(n) for n in GOLD["names"]]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_resize_equals_pillow_golden(name):
    want = GOLD[name + "/out"]
    got = P.resize_bilinear_u8(GOLD[name + "/in"], want.shape[0], want.shape[1])
    assert got.dtype == np.uint8 and np.array_equal(got, want)


def test_oracle_normalize_equals_torch_golden():
    assert np.array_equal(P.to_tensor_normalize(GOLD["plan_crop/out"]), GOLD["plan_crop/normalized"])


def test_oracle_against_live_pillow_if_present():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    for (h, w, H, W) in [(61, 47, 32, 32), (20, 31, 48, 64), (128, 128, 128, 64)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(a).resize((W, H), Image.BILINEAR))
        assert np.array_equal(P.resize_bilinear_u8(a, H, W), want)


@pytest.mark.parametrize("sizes", [(1105, 512), (800, 512), (40, 64), (512, 512), (5, 32), (300, 32), (7, 7)])
def test_library_coefficient_tables_equal_the_oracle(sizes):
    a, b = sizes
    ks = L.call("fea_resize_ksize", a, b)
    bounds = torch.empty(2 * b, dtype=torch.int32)
    coef = torch.empty(b * ks, dtype=torch.int32)
    assert L.call("fea_resize_coeffs", a, b, bounds, coef) == ks
    k2, b2, c2 = P.resize_coeffs(a, b)
    assert ks == k2
    assert np.array_equal(bounds.numpy().reshape(-1, 2), b2) and np.array_equal(coef.numpy().reshape(b, ks), c2)


def test_library_lut_is_totensor_normalize():
    lut = torch.empty(768)
    L.call("fea_normalize_lut", torch.tensor(P.IMAGENET_MEAN), torch.tensor(P.IMAGENET_STD), 3, lut)
    v = np.arange(256, dtype=np.uint8)[None, :, None].repeat(3, axis=2)
    assert np.array_equal(P.to_tensor_normalize(v).reshape(3, 256), lut.numpy().reshape(3, 256))


def test_mask_target_threshold_edge():
    g = np.array([[126, 127], [128, 255]], np.uint8)
    assert P.mask_target(g, 2).tolist() == [[0, 0], [1, 1]]
