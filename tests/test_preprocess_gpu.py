"""HIP input pipeline (fea_resize_u8 through fea_amd.data) against the CPU oracle and the
Pillow-made golden vectors: integer/byte work and a table lookup, so every comparison is
bit-exact (custom_dataset.py:23-28)."""
import os

import numpy as np
import pytest
import torch

from fea_amd import _lib as L
from fea_amd.data import FloorPlanPairs, ImagePreprocessor
from oracle import preprocess_oracle as P

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_golden.npz"))
NAMES = [str(n) for n in GOLD["names"]]


@pytest.mark.parametrize("name", NAMES)
def test_resize_equals_pillow_golden(name):
    want = GOLD[name + "/out"]
    pre = ImagePreprocessor(64)
    got = pre.resize(torch.from_numpy(GOLD[name + "/in"]), want.shape[0], want.shape[1]).cpu().numpy()
    assert np.array_equal(got, want)


def test_full_transform_golden_and_oracle():
    pre = ImagePreprocessor(96)
    got = pre.image(torch.from_numpy(GOLD["plan_crop/in"])).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, GOLD["plan_crop/normalized"])
    # the shipped data's shape: 1105x800 (and a portrait one) -> 512x512, random content
    rng = np.random.default_rng(9)
    pre = ImagePreprocessor(512)
    for (h, w) in [(800, 1105), (1200, 791), (512, 512), (300, 2000)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(pre.image(torch.from_numpy(a)).cpu().numpy(), P.transform(a, 512)), (h, w)
        m = a[:, :, 0]
        assert np.array_equal(pre.mask(torch.from_numpy(m)).cpu().numpy(), P.mask_target(m, 512)), (h, w)


def test_extreme_sizes_and_constant_images():
    pre = ImagePreprocessor(16)
    for (h, w) in [(1, 1), (1, 37), (16, 16), (4000, 3)]:
        for fill in (0, 255):
            a = np.full((h, w, 3), fill, np.uint8)
            got = pre.resize(torch.from_numpy(a), 16, 16).cpu().numpy()
            assert np.array_equal(got, P.resize_bilinear_u8(a, 16, 16)) and (got == fill).all()


def test_bad_arguments_raise():
    pre = ImagePreprocessor(32)
    with pytest.raises(ValueError):
        pre.image(torch.zeros(8, 8, dtype=torch.uint8))
    with pytest.raises(ValueError):
        pre.image(torch.zeros(8, 8, 3))
    with pytest.raises(L.FeaError):
        pre._run(torch.zeros(8, 8, 2, dtype=torch.uint8), torch.empty(32, 32, 2, dtype=torch.uint8, device="cuda"), 0, 32, 32)
    with pytest.raises(L.FeaError):
        ImagePreprocessor(32, device="cpu")


def test_pair_loader_matches_the_reference_transform(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(4)
    fp, mk = tmp_path / "floor_plans2", tmp_path / "dataset_masks"
    fp.mkdir(); mk.mkdir()
    srcs = {}
    for i in range(5):
        a = rng.integers(0, 256, (70 + 3 * i, 90 - 2 * i, 3), dtype=np.uint8)
        name = f"{1000 + i}abc_floorplan_1.png"
        Image.fromarray(a).convert("P").save(fp / name)            # palette image, as the shipped data
        g = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        Image.fromarray(g).save(mk / name)
        srcs[name] = None
    (fp / "unpaired.png").write_bytes((fp / name).read_bytes())
    ds = FloorPlanPairs(str(fp), str(mk), img_size=32, num_classes=6, workers=2)
    assert len(ds) == 5
    seen = 0
    for rank in (0, 1):
        for x, t in ds.batches(2, shuffle=True, seed=1, rank=rank, world=2):
            assert x.shape[1:] == (3, 32, 32) and t.shape[1:] == (6, 32, 32) and x.is_cuda
            seen += x.shape[0]
    assert seen == 5
    x, t = next(ds.batches(5, shuffle=False))
    for j, (pi, pm) in enumerate(ds.items):
        a = np.asarray(Image.open(pi).convert("RGB"))
        g = np.asarray(Image.open(pm).convert("L"))
        assert np.array_equal(x[j].cpu().numpy(), P.transform(a, 32))
        assert np.array_equal(t[j, 0].cpu().numpy(), P.mask_target(g, 32))
        assert int(t[j, 1:].sum()) == 0
