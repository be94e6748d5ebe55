"""Window geometry of the tiled inference path (BASELINE.json configs[4]); no GPU needed."""
import pytest

from fea_amd.inference import HALO, window_plan


@pytest.mark.parametrize("H,W,core", [(2048, 2048, 512), (2048, 2048, 1024), (384, 384, 64), (1024, 1536, 256), (512, 512, 512)])
def test_cores_partition_the_image_and_keep_their_halo(H, W, core):
    win, plan = window_plan(H, W, core)
    cover = [[0] * (W // 16) for _ in range(H // 16)]
    for wy, wx, cy0, cy1, cx0, cx1 in plan:
        if len(plan) > 1:
            assert wy % 16 == 0 and wx % 16 == 0 and 0 <= wy <= H - win and 0 <= wx <= W - win
            # distance from the core to every window edge that is not an image edge
            assert wy == 0 or cy0 - wy >= HALO
            assert wx == 0 or cx0 - wx >= HALO
            assert wy + win == H or wy + win - cy1 >= HALO
            assert wx + win == W or wx + win - cx1 >= HALO
            assert wy <= cy0 and cy1 <= wy + win and wx <= cx0 and cx1 <= wx + win
        for y in range(cy0 // 16, cy1 // 16):
            for x in range(cx0 // 16, cx1 // 16):
                cover[y][x] += 1
    assert all(v == 1 for row in cover for v in row)


def test_rejects_sizes_the_unet_cannot_take():
    with pytest.raises(ValueError):
        window_plan(250, 256, 64)
