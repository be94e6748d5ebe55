"""Host-side logic that needs no GPU: the drop-in module's construction,
state_dict contract and error behaviour, and the gradient-bucket planning."""
import pytest
import torch

import fea_amd
from fea_amd import _lib as L
from fea_amd.engine import UNetEngine, param_order
from tests.helpers import digest_err, load_golden

GOLD = load_golden()


@pytest.mark.parametrize("name", ["train_32", "train_c1"])
def test_same_seed_gives_the_reference_weights(name):
    """Construction order and default init equal the reference's
    (models.py:10-23): same seed => same state_dict, key for key."""
    case = GOLD[name]
    torch.manual_seed(case["seed"])
    m = fea_amd.FloorPlanSegmentationModel(in_channels=case["in_channels"], num_classes=case["num_classes"])
    sd = m.state_dict()
    assert list(sd.keys()) == list(case["init"].keys())
    for k, g in case["init"].items():
        assert digest_err(g, sd[k].float()) < 1e-6, k


def test_parameter_count_and_attribute_names():
    m = fea_amd.FloorPlanSegmentationModel()
    assert sum(p.numel() for p in m.parameters()) == 31_043_846
    for attr in ["enc1", "enc2", "enc3", "enc4", "bottleneck", "upconv4", "dec4", "upconv3", "dec3",
                 "upconv2", "dec2", "upconv1", "dec1", "out", "pool"]:
        assert hasattr(m, attr), attr
    assert m.upconv1.weight.shape == (128, 64, 2, 2) and m.out.weight.shape == (6, 64, 1, 1)


def test_load_state_dict_accepts_reference_checkpoint_spellings():
    src = fea_amd.FloorPlanSegmentationModel()
    sd = src.state_dict()
    for prefix in ("", "module.", "segmentation_model.", "module.segmentation_model."):
        dst = fea_amd.FloorPlanSegmentationModel()
        ck = {prefix + k: v.clone() for k, v in sd.items()}
        ck["module.fusion_model.some.weight" if prefix else "unrelated"] = torch.zeros(1)
        if not prefix:
            ck.pop("unrelated")
        dst.load_state_dict(ck)
        for k, v in dst.state_dict().items():
            assert torch.equal(v, sd[k]), (prefix, k)
    with pytest.raises(RuntimeError):
        fea_amd.FloorPlanSegmentationModel().load_state_dict({"enc1.0.weight": torch.zeros(64, 3, 3, 3)})


def test_cpu_input_raises_instead_of_falling_back():
    m = fea_amd.FloorPlanSegmentationModel()
    with pytest.raises(L.FeaError):
        m(torch.zeros(1, 3, 16, 16))


def test_compute_dtype_knob():
    assert fea_amd.FloorPlanSegmentationModel().compute_dtype == torch.float32      # reference behaviour
    assert fea_amd.FloorPlanSegmentationModel(compute_dtype="bf16").compute_dtype == torch.bfloat16
    with pytest.raises(ValueError):
        fea_amd.FloorPlanSegmentationModel(compute_dtype="fp8")


def test_flat_gradient_layout_is_reverse_forward_and_complete():
    m = fea_amd.FloorPlanSegmentationModel()
    order = param_order(m)
    assert set(order) == {n for n, _ in m.named_parameters()} and len(order) == len(set(order)) == 82
    assert order[:2] == ["out.weight", "out.bias"] and order[-1] == "enc1.1.bias"
    eng = UNetEngine(m, torch.float32)
    eng._ensure_flat_grad()
    prev_end = 0
    for n in order:
        off, numel = eng.grad_offsets[n]
        assert off >= prev_end and off % 64 == 0
        prev_end = off + numel
        assert eng.grad_views[n].shape == dict(m.named_parameters())[n].shape
    assert eng.flat_grad.numel() >= 31_043_846


def test_product_synthetic_batch_is_the_oracles():
    """bench.py / smoke() take their inputs from fea_amd.data.synthetic_batch; the CPU baseline
    takes the oracle's: they must be the same tensors."""
    from fea_amd.data import synthetic_batch
    from oracle import unet_oracle as O
    a, ta = synthetic_batch(2, 48, 32, seed=77)
    b, tb = O.synthetic_batch(2, 48, 32, seed=77)
    assert torch.equal(a, b) and torch.equal(ta, tb)


def test_trace_is_free_when_off(monkeypatch):
    """fea_amd.trace (roctx ranges): with FEA_AMD_ROCTX unset the decorator returns the function it was given and no
    marker library is loaded."""
    import importlib
    monkeypatch.delenv("FEA_AMD_ROCTX", raising=False)
    from fea_amd import trace
    trace = importlib.reload(trace)
    assert trace.LEVEL == 0 and not trace.CALLS

    def f(x):
        return x + 1
    assert trace.traced("anything")(f) is f
    with trace.span("nothing happens"):
        assert f(1) == 2
