"""The CPU oracle (oracle/unet_oracle.py) against golden digests produced by
the REFERENCE module (oracle/make_golden.py).  CPU only."""
from collections import OrderedDict

import pytest
import torch

from oracle import unet_oracle as O
from tests.helpers import digest_err, load_golden

GOLD = load_golden()
TOL = 1e-4  # north_star: fp32 forward/backward within 1e-4 relative


def case_inputs(case):
    """Mirror of oracle/make_golden.py:make_inputs."""
    x, tgt = O.synthetic_batch(case["n"], case["h"], case["w"], num_classes=case["num_classes"],
                               seed=case["seed"] + 100, in_channels=case["in_channels"])
    if case["kind"] == "randn":
        g = torch.Generator().manual_seed(case["seed"] + 200)
        x = torch.randn(case["n"], case["in_channels"], case["h"], case["w"], generator=g)
    return x, tgt


def _case_state(case):
    sd = O.init_state_dict(case["seed"], case.get("in_channels", 3), case.get("num_classes", 6))
    return sd


@pytest.mark.parametrize("name", ["train_32", "train_64", "train_c1"])
def test_init_matches_reference(name):
    case = GOLD[name]
    sd = _case_state(case)
    assert list(sd.keys()) == list(case["init"].keys())
    for k, g in case["init"].items():
        assert digest_err(g, sd[k].float()) < 1e-6, k
    n_params = sum(v.numel() for k, v in sd.items() if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    if name == "train_32":
        assert n_params == 31_043_846


@pytest.mark.parametrize("name", ["train_32", "train_64", "train_c1", "plan_64"])
def test_train_steps_match_reference(name):
    case = GOLD[name]
    torch.set_num_threads(8)
    sd = _case_state(case)
    params, bufs = O.split_state(sd)
    x, tgt = case_inputs(case)
    assert digest_err(case["x"], x) < 1e-7
    assert int(tgt.sum()) == case["target_sum"]
    state = {}
    for rec in case["train"]:
        loss, logits, grads, nb = O.loss_and_grads(params, bufs, x, tgt, training=True)
        # non-strict cases: after the first optimizer step the parameters carry
        # the (Adam-amplified) effect of kink flips, so later steps are loose
        ftol = TOL if (case["strict"] or rec["step"] == 0) else 2e-2
        assert abs(float(loss) - rec["loss"]) <= ftol * abs(rec["loss"]), (rec["step"], float(loss), rec["loss"])
        assert digest_err(rec["logits"], logits) < ftol, rec["step"]
        if "grads" in rec:
            worst = max((digest_err(g, grads[k]), k) for k, g in rec["grads"].items()
                        if not _bn_cancelled_bias(k))
            # ReLU / max-pool kinks: an activation within fp32 rounding of 0
            # (or a pool tie) flips between two fp32 implementations and moves
            # a layer's gradient by ~1/sqrt(numel).  strict cases were chosen
            # (oracle/make_golden.py:well_conditioned) to have no such flip.
            assert worst[0] < (5e-4 if case["strict"] else 5e-2), worst
            for k, g in rec["buffers"].items():
                assert digest_err(g, nb[k].float()) < TOL, k
        clipped, norm = O.clip_grad_norm(grads, 1.0)
        assert abs(float(norm) - rec["grad_norm"]) <= 2e-4 * rec["grad_norm"]
        params = O.adamw_step(params, clipped, state)
        bufs = nb
        if "params_after" in rec:
            # Post-AdamW parameters are compared on the strided samples with
            # an absolute bound: an Adam update is lr*g/(|g|+eps)-like, so (a)
            # BN-cancelled conv biases (noise gradients) move by +-lr with a
            # random sign and (b) elements with |g| ~ eps are ill-conditioned.
            lr, nstep = 1e-4, rec["step"] + 1
            for k, g in rec["params_after"].items():
                ref = torch.tensor(g["samples"], dtype=torch.float64)
                got = torch.tensor(O.digest(params[k], len(g["samples"]))["samples"], dtype=torch.float64)
                err = (ref - got).abs().max().item()
                if _bn_cancelled_bias(k) or not case["strict"]:
                    assert err <= 2.1 * lr * nstep, (k, err)
                else:
                    assert err <= TOL * ref.abs().max().item() + 0.1 * lr * nstep, (k, err)
    logits, _ = O.forward(params, bufs, x, training=False)
    assert digest_err(case["eval_logits"], logits) < (5e-4 if case["strict"] else 5e-2)
    pred, inter, union = O.mask_iou_counts(logits, tgt)
    if not case["strict"]:
        return
    # masks are exact GIVEN identical logits; across two fp32 implementations a
    # pixel with |logit| ~ 1e-7 may flip, so allow a handful.
    assert abs(int(pred.sum()) - case["eval_iou"]["pred_sum"]) <= 4
    for a, b in zip(inter.tolist(), case["eval_iou"]["inter"]):
        assert abs(a - b) <= 4
    for a, b in zip(union.tolist(), case["eval_iou"]["union"]):
        assert abs(a - b) <= 4


def _bn_cancelled_bias(k):
    """Conv biases that feed a train-mode BatchNorm have a mathematically zero
    gradient (the reference's value is fp32 rounding noise), so they are not
    compared by relative error."""
    parts = k.split(".")
    return len(parts) == 3 and parts[1] in ("0", "3") and parts[2] == "bias"


def test_dlogits_backward_matches_reference():
    case = GOLD["dlogits_32"]
    sd = O.init_state_dict(case["seed"])
    params, bufs = O.split_state(sd)
    g = torch.Generator().manual_seed(case["seed"] + 7)
    x = torch.randn(case["n"], 3, case["h"], case["w"], generator=g)
    dl = torch.randn(case["n"], 6, case["h"], case["w"], generator=g) / (case["n"] * case["h"] * case["w"])
    assert digest_err(case["x"], x) < 1e-7 and digest_err(case["dlogits"], dl) < 1e-7
    logits, grads, _ = O.grads_from_dlogits(params, bufs, x, dl)
    assert digest_err(case["logits"], logits) < TOL
    worst = max((digest_err(gd, grads[k]), k) for k, gd in case["grads"].items() if not _bn_cancelled_bias(k))
    assert worst[0] < 5e-4, worst


def test_config1_eval_forward():
    case = GOLD["config1"]
    sd = O.init_state_dict(0)
    params, bufs = O.split_state(sd)
    x, _ = O.synthetic_batch(1, 256, 256, seed=4321)
    assert digest_err(case["x"], x) < 1e-7
    with torch.no_grad():
        y, _ = O.forward(params, bufs, x, training=False)
    assert digest_err(case["logits"], y) < TOL
    assert abs(int((y > 0).sum()) - case["pred_sum"]) <= 4


def test_cancelled_bias_grads_are_noise():
    """Documents the exclusion above: those gradients are ~1e-7 of the BN
    beta gradient scale in the reference."""
    case = GOLD["train_32"]
    g0 = case["train"][0]["grads"]
    assert g0["enc1.0.bias"]["l2"] < 1e-4 * max(g0["enc1.1.bias"]["l2"], 1e-12) + 1e-7


def test_cosine_lr_matches_torch():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=1e-6)
    for e in range(10):
        assert abs(opt.param_groups[0]["lr"] - O.cosine_lr(1e-4, e, 50)) < 1e-12
        opt.step()
        sch.step()


def test_config1_on_a_real_floor_plan_matches_the_reference():
    """BASELINE configs[0] as worded: the reference UNet, fp32 eval, batch 1, 256x256, on an image of the reference's
    floor_plans2/ (tests/golden/config1_real.*, made by oracle/make_golden_config1_real.py from the reference module
    and Pillow).  Pins both oracles on real data: the input-pipeline restatement bit for bit, the UNet to 1e-4."""
    import json
    import os
    import numpy as np
    from oracle import preprocess_oracle as P
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rec = json.load(open(os.path.join(gdir, "config1_real.json")))
    data = np.load(os.path.join(gdir, "config1_real.npz"))
    assert list(data["image"].shape) == rec["image_shape"]
    small = P.resize_bilinear_u8(data["image"], rec["h"], rec["w"])
    assert np.array_equal(small, data["resized"])                    # == Pillow's output, bit for bit
    x = torch.from_numpy(P.to_tensor_normalize(small)).unsqueeze(0)
    assert digest_err(rec["x"], x) < 1e-7
    params, bufs = O.split_state(O.init_state_dict(rec["seed"]))
    with torch.no_grad():
        y, _ = O.forward(params, bufs, x, training=False)
    assert digest_err(rec["logits"], y) < TOL
    assert abs(int((y > 0).sum()) - rec["pred_sum"]) <= 4


def test_oracle_against_reference_dense_tensors():
    """Element by element, not through digests: the oracle against the reference's FULL logits, gradients (whole
    tensors up to 65,536 elements, 8,192 elements of each larger one), BatchNorm buffers, post-AdamW parameters and
    eval logits of case train_32 (tests/golden/unet_dense.npz, written by oracle/make_golden_dense.py from the
    reference module).  Bound: max element error <= 1e-4 of the tensor's RMS (train_32 is a well-conditioned seed: no
    ReLU / pool decision within rounding of its kink)."""
    from tests.helpers import dense_err, dense_take, load_dense
    z, meta = load_dense()
    case = GOLD["train_32"]
    assert (meta["seed"], meta["n"], meta["h"], meta["w"]) == (case["seed"], case["n"], case["h"], case["w"])
    torch.set_num_threads(8)
    params, bufs = O.split_state(_case_state(case))
    x, tgt = case_inputs(case)
    loss, logits, grads, nb = O.loss_and_grads(params, bufs, x, tgt, training=True)
    assert abs(float(loss) - float(z["loss"])) <= 1e-6 * abs(float(z["loss"]))
    assert dense_err(logits, z["logits"]) < 1e-5
    worst = max((dense_err(dense_take(g, meta), z["grad/" + k]), k) for k, g in grads.items() if not _bn_cancelled_bias(k))
    assert worst[0] < 1e-4, worst
    for k, b in nb.items():
        if not k.endswith("num_batches_tracked"):
            assert dense_err(dense_take(b, meta), z["buffer/" + k]) < 1e-5, k
    clipped, _ = O.clip_grad_norm(grads, 1.0)
    after = O.adamw_step(params, clipped, {})
    for k, p in after.items():
        if not _bn_cancelled_bias(k):
            ref = torch.as_tensor(z["after/" + k]).double()
            # an Adam step moves an element by <= lr; elements whose gradient is ~eps are sign-like: absolute bound
            assert (dense_take(p, meta).double() - ref).abs().max().item() <= 1e-6 * ref.abs().max().item() + 0.1 * 1e-4, k
    le, _ = O.forward(after, nb, x, training=False)
    # eval mode uses the running statistics, so the +-lr sign-like moves of the BatchNorm-cancelled conv biases (noise
    # gradients) no longer cancel: same 5e-4 as the digest test above
    assert dense_err(le, z["eval_logits"]) < 5e-4
