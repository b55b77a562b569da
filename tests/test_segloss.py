"""Multi-scale segmentation loss.  PARITY UNPINNED by reference outputs (the reference class cannot be imported
here: oracle/segloss_oracle.py header); pinned instead by known answers, by torch's own ops and by torch autograd.
CPU: the oracle's known answers, the module's host path == the oracle (both modes), reference interface.
GPU: the HIP entry points (through the module and the C ABI) == the oracle, forward and gradient, incl. nearest target resize."""
import math

import pytest
import torch

from oracle import segloss_oracle as O


def _data(B=3, sizes=((16, 16), (8, 8), (4, 4)), tsize=None, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    preds, tg = {}, []
    for k, (H, W) in zip(("p3", "p4", "p5"), sizes):
        preds[k] = (torch.randn(B, 1, H, W, generator=g) * 2).to(dtype)
        th, tw = (H, W) if tsize is None else tsize
        tg.append((torch.rand(B, 1, th, tw, generator=g) > 0.7).float())
    return preds, tg


def test_known_answers():
    """logits 0, target 0: bce = ln 2 and dice = 1 - s/(N/2 + s); perfect confident prediction -> both terms ~0."""
    N = 8 * 8
    preds = {"p3": torch.zeros(2, 1, 8, 8)}
    total, logs = O.forward(preds, [torch.zeros(2, 1, 8, 8)], O.SegLossConfig())
    assert abs(logs["p3_bce"] - math.log(2)) < 1e-6
    assert abs(logs["p3_dice"] - (1 - 1.0 / (N / 2 + 1.0))) < 1e-6
    assert abs(float(total) - (logs["p3_bce"] + logs["p3_dice"])) < 1e-6
    t = (torch.rand(2, 1, 8, 8, generator=torch.Generator().manual_seed(1)) > 0.5).float()
    total, logs = O.forward({"p3": (t * 2 - 1) * 30}, [t], O.SegLossConfig())
    assert logs["p3_bce"] < 1e-6 and logs["p3_dice"] < 1e-6
    # scale weights, lambda, missing levels, 3-D targets
    preds, tg = _data()
    cfg = O.SegLossConfig(scale_weights=(0.5, 2.0, 3.0), loss_lambda=0.25, bce_weight=0.7, dice_weight=1.3)
    total, logs = O.forward(preds, [t_.squeeze(1) for t_ in tg], cfg)
    want = 0.25 * sum(w * (0.7 * logs[f"{k}_bce"] + 1.3 * logs[f"{k}_dice"]) for k, w in zip(("p3", "p4", "p5"), (0.5, 2.0, 3.0)))
    assert abs(float(total) - want) < 1e-5
    total2, logs2 = O.forward({"p4": preds["p4"]}, tg, O.SegLossConfig())
    assert set(logs2) == {"p4_bce", "p4_dice", "p4_combined", "seg_total"}
    assert O.forward(preds, tg, O.SegLossConfig(enabled=False)) == (torch.zeros(()), {})


@pytest.mark.parametrize("ufl", [False, True])
@pytest.mark.parametrize("tsize", [None, (64, 64), (17, 23)])
def test_module_host_path_equals_oracle(ufl, tsize):
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    preds, tg = _data(tsize=tsize, seed=3)
    kw = dict(scale_weights=(1.0, 0.5, 0.25), loss_lambda=0.8, use_unified_focal=ufl)
    po = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    pm = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    to, lo = O.forward(po, tg, O.SegLossConfig(**kw))
    tm, lm = SegmentationLoss(SegLossConfig(**kw))(pm, tg)
    assert lo.keys() == lm.keys() and all(abs(lo[k] - lm[k]) < 1e-6 for k in lo)
    to.backward(); tm.backward()
    for k in preds:
        assert torch.allclose(po[k].grad, pm[k].grad, rtol=1e-5, atol=1e-8)


def test_non_finite_raises_like_the_reference():
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    bad = {"p3": torch.full((1, 1, 4, 4), float("nan"))}
    with pytest.raises(FloatingPointError):
        SegmentationLoss(SegLossConfig())(bad, [torch.zeros(1, 1, 4, 4)])
    with pytest.raises(FloatingPointError):
        O.forward(bad, [torch.zeros(1, 1, 4, 4)], O.SegLossConfig())


# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("B,sizes,tsize", [
    (32, ((80, 80), (40, 40), (20, 20)), None),            # BASELINE configs[1] mask sizes
    (3, ((16, 16), (8, 8), (4, 4)), (64, 64)),             # nearest gather from a full-resolution target
    (2, ((17, 23), (9, 12), (5, 6)), (68, 92)),            # odd sizes, non-integer ratios on one axis
    (1, ((1, 1), (3, 2), (2, 5)), None),
])
def test_device_loss_and_gradient_match_the_oracle(built_lib, B, sizes, tsize):
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    preds, tg = _data(B=B, sizes=sizes, tsize=tsize, seed=11)
    kw = dict(scale_weights=(1.0, 0.5, 2.0), loss_lambda=0.7, bce_weight=0.9, dice_weight=1.1, smooth=1.0)
    po = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    to, lo = O.forward(po, tg, O.SegLossConfig(**kw))
    (to * 1.7).backward()
    pd = {k: v.cuda().requires_grad_(True) for k, v in preds.items()}
    td, ld = SegmentationLoss(SegLossConfig(**kw))(pd, [t.cuda() for t in tg])
    (td * 1.7).backward()
    assert ld.keys() == lo.keys()
    for k in lo:
        assert abs(ld[k] - lo[k]) <= 1e-5 * max(1.0, abs(lo[k])), (k, ld[k], lo[k])
    assert abs(float(td.detach()) - float(to.detach())) <= 1e-5 * max(1.0, abs(float(to.detach())))
    for k in preds:
        g, w = pd[k].grad.cpu(), po[k].grad
        assert float((g - w).abs().max()) <= 1e-4 * float(w.abs().max()) + 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
def test_device_half_precision_logits(built_lib, dtype, tol):
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    preds, tg = _data(B=4, seed=5, dtype=dtype)
    po = {k: v.float().requires_grad_(True) for k, v in preds.items()}
    to, _ = O.forward(po, tg, O.SegLossConfig())
    to.backward()
    pd = {k: v.cuda().requires_grad_(True) for k, v in preds.items()}
    td, _ = SegmentationLoss(SegLossConfig())(pd, [t.cuda() for t in tg])
    td.backward()
    assert abs(float(td.detach()) - float(to.detach())) < 1e-5 * abs(float(to.detach())) + 1e-6   # the sums are fp32 either way
    for k in preds:
        assert pd[k].grad.dtype == dtype
        w = po[k].grad
        assert float((pd[k].grad.float().cpu() - w).abs().max()) <= tol * float(w.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("tsize,gamma,delta", [(None, 0.5, 0.6), ((64, 64), 0.75, 0.3), (None, 0.2, 0.9)])
def test_device_unified_focal_mode(built_lib, tsize, gamma, delta):
    """_lmf / _lmft on the device (pow, clamps and their gradient gates) == the oracle (torch ops + autograd)."""
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    preds, tg = _data(B=4, tsize=tsize, seed=2)
    preds["p4"][0] = 30.0                                   # saturated logits: pt / base clamps active
    preds["p4"][1] = -30.0
    kw = dict(use_unified_focal=True, ufl_gamma=gamma, ufl_delta=delta, ufl_lambda=0.4, scale_weights=(1.0, 2.0, 0.5), loss_lambda=1.3)
    po = {k: v.clone().requires_grad_(True) for k, v in preds.items()}
    to, lo = O.forward(po, tg, O.SegLossConfig(**kw))
    to.backward()
    pd = {k: v.cuda().requires_grad_(True) for k, v in preds.items()}
    td, ld = SegmentationLoss(SegLossConfig(**kw))(pd, [t.cuda() for t in tg])
    td.backward()
    for k in lo:
        assert abs(ld[k] - lo[k]) <= 1e-5 * max(1.0, abs(lo[k])), (k, ld[k], lo[k])
    for k in preds:
        g, w = pd[k].grad.cpu(), po[k].grad
        assert float((g - w).abs().max()) <= 1e-4 * float(w.abs().max()) + 1e-9, k


@pytest.mark.gpu
def test_device_bad_shapes(built_lib):
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    with pytest.raises(RuntimeError):
        SegmentationLoss(SegLossConfig())({"p3": torch.zeros(2, 3, 4, 4).cuda()}, [torch.zeros(2, 1, 4, 4).cuda()])
