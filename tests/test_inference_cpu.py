"""Window geometry of the tiled inference path (BASELINE.json configs[4]); no GPU needed."""
import pytest

from fea_amd.inference import HALO, window_plan


@pytest.mark.parametrize("H,W,core", [(2048, 2048, 512), (2048, 2048, 1024), (384, 384, 64), (1024, 1536, 256), (512, 512, 512),
                                      (2048, 2048, (1024, 2048)), (2048, 2048, (2048, 512)), (640, 384, (128, 64)), (256, 1024, 256)])
def test_cores_partition_the_image_and_keep_their_halo(H, W, core):
    wh, ww, plan = window_plan(H, W, core)
    assert wh % 16 == 0 and ww % 16 == 0 and wh <= H and ww <= W
    cover = [[0] * (W // 16) for _ in range(H // 16)]
    for wy, wx, cy0, cy1, cx0, cx1 in plan:
        assert wy % 16 == 0 and wx % 16 == 0 and 0 <= wy <= H - wh and 0 <= wx <= W - ww
        # distance from the core to every window edge that is not an image edge
        assert wy == 0 or cy0 - wy >= HALO
        assert wx == 0 or cx0 - wx >= HALO
        assert wy + wh == H or wy + wh - cy1 >= HALO
        assert wx + ww == W or wx + ww - cx1 >= HALO
        assert wy <= cy0 and cy1 <= wy + wh and wx <= cx0 and cx1 <= wx + ww
        for y in range(cy0 // 16, cy1 // 16):
            for x in range(cx0 // 16, cx1 // 16):
                cover[y][x] += 1
    assert all(v == 1 for row in cover for v in row)


def test_a_direction_the_window_covers_is_not_cut():
    wh, ww, plan = window_plan(2048, 2048, (1024, 2048))
    assert (wh, ww) == (1216, 2048) and len(plan) == 2
    wh, ww, plan = window_plan(512, 512, 512)
    assert (wh, ww) == (512, 512) and plan == [(0, 0, 0, 512, 0, 512)]


def test_rejects_sizes_the_unet_cannot_take():
    with pytest.raises(ValueError):
        window_plan(250, 256, 64)
