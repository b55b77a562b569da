"""Multi-scale segmentation loss.  Pinned by outputs of the reference's own SegmentationLoss / MGAModel.loss
(tests/golden/segloss_*.npz, kendall_*.npz, written by oracle/gen_golden_segloss.py in the build container).
CPU: the oracle and the module's host path vs those goldens, known answers, reference interface.
GPU: the HIP entry points (through the module and the C ABI) vs the goldens and the oracle, forward and gradient, incl. nearest and
bilinear target resize and the Kendall combine."""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import segloss_oracle as O


def seg_golden_names():
    return sorted(f[len("segloss_"):-4] for f in os.listdir(GOLDEN) if f.startswith("segloss_") and f.endswith(".npz"))


def load_seg_golden(name):
    z = np.load(os.path.join(GOLDEN, f"segloss_{name}.npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    preds = {k[len("logits."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("logits.")}
    grads = {k[len("grad."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("grad.")}
    targets = [torch.from_numpy(z[f"target.{i}"]) for i in range(sum(1 for k in z.files if k.startswith("target.")))]
    order = [k for k in ("p3", "p4", "p5") if k in preds]
    return dict(preds={k: preds[k] for k in order}, grads=grads, targets=targets, meta=meta)


def _check_against_golden(d, total, logs, grads, tol_loss=1e-5, tol_grad=1e-4):
    want = d["meta"]["logs"]
    assert set(logs) == set(want), (sorted(logs), sorted(want))
    for k, v in want.items():
        assert abs(logs[k] - v) <= tol_loss * max(1.0, abs(v)), (k, logs[k], v)
    assert abs(float(total) - d["meta"]["total"]) <= tol_loss * max(1.0, abs(d["meta"]["total"]))
    for k, w in d["grads"].items():
        g = grads[k]
        g = torch.zeros_like(w) if g is None else g.detach().float().cpu()
        assert float((g - w).abs().max()) <= tol_grad * float(w.abs().max()) + 1e-9, k


@pytest.mark.parametrize("name", seg_golden_names())
def test_oracle_matches_the_reference_golden(name):
    d = load_seg_golden(name)
    cfg = O.SegLossConfig(**d["meta"]["cfg"])
    leaf = {k: v.float().clone().requires_grad_(True) for k, v in d["preds"].items()}
    total, logs = O.forward(leaf, d["targets"], cfg, bilinear_targets=d["meta"]["prob_mode"])
    total.backward()
    _check_against_golden(d, total.detach(), logs, {k: v.grad for k, v in leaf.items()}, tol_loss=1e-6, tol_grad=1e-6)


@pytest.mark.parametrize("name", seg_golden_names())
def test_module_host_path_matches_the_reference_golden(name, monkeypatch):
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    d = load_seg_golden(name)
    if d["meta"]["prob_mode"]:
        monkeypatch.setenv("MGA_PROB_MODE", "1")
    else:
        monkeypatch.delenv("MGA_PROB_MODE", raising=False)
    leaf = {k: v.float().clone().requires_grad_(True) for k, v in d["preds"].items()}
    total, logs = SegmentationLoss(SegLossConfig(**d["meta"]["cfg"]))(leaf, d["targets"])
    total.backward()
    _check_against_golden(d, total.detach(), logs, {k: v.grad for k, v in leaf.items()}, tol_loss=1e-6, tol_grad=1e-6)


@pytest.mark.parametrize("tag", ["init", "trained"])
def test_kendall_combine_matches_the_reference_model(tag):
    """MGAModel.loss's multi-task combine as the reference model itself evaluated it (model/model.py:204-206)."""
    from mga_yolo_amd import kendall_combine
    z = np.load(os.path.join(GOLDEN, f"kendall_{tag}.npz"))
    for fn in (O.kendall_combine, kendall_combine):
        det = torch.from_numpy(z["det_loss"]).clone().requires_grad_(True)
        seg = torch.from_numpy(z["seg_total"]).clone().requires_grad_(True)
        lv = torch.from_numpy(z["log_vars"]).clone().requires_grad_(True)
        total = fn(det, seg, lv)
        assert torch.allclose(total.detach(), torch.from_numpy(z["total"]), rtol=1e-6, atol=1e-6)
        total.sum().backward()
        assert torch.allclose(lv.grad, torch.from_numpy(z["g_log_vars"]), rtol=1e-5, atol=1e-5)
        assert torch.allclose(det.grad, torch.exp(-lv.detach()[0]).expand(3), rtol=1e-6)


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


@pytest.mark.gpu
@pytest.mark.parametrize("name", seg_golden_names())
def test_device_path_matches_the_reference_golden(built_lib, name, monkeypatch):
    """The HIP loss (through the module: C ABI mgaseg_*) against what the reference's own SegmentationLoss produced."""
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    d = load_seg_golden(name)
    if d["meta"]["prob_mode"]:
        monkeypatch.setenv("MGA_PROB_MODE", "1")
    else:
        monkeypatch.delenv("MGA_PROB_MODE", raising=False)
    half = d["meta"]["dtype"] == "float16"
    leaf = {k: v.cuda().requires_grad_(True) for k, v in d["preds"].items()}
    total, logs = SegmentationLoss(SegLossConfig(**d["meta"]["cfg"]))(leaf, [t.cuda() for t in d["targets"]])
    total.backward()
    _check_against_golden(d, total.detach().cpu(), logs, {k: v.grad for k, v in leaf.items()}, tol_loss=2e-5, tol_grad=2e-3 if half else 1e-4)


@pytest.mark.gpu
def test_device_mixed_dtypes_stay_on_the_device(built_lib):
    """Levels of different element types are computed in fp32 ON THE DEVICE (no torch-op fallback for device tensors)."""
    from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss
    preds, tg = _data(B=2, seed=4)
    pd = {"p3": preds["p3"].cuda().half().requires_grad_(True), "p4": preds["p4"].cuda().requires_grad_(True)}
    crit = SegmentationLoss(SegLossConfig())
    called = []
    orig = crit._torch_forward
    crit._torch_forward = lambda *a, **k: called.append(1) or orig(*a, **k)
    total, logs = crit(pd, [t.cuda() for t in tg])
    total.backward()
    assert not called and pd["p3"].grad.dtype == torch.float16 and pd["p4"].grad.dtype == torch.float32
    po = {k: v.detach().float().cpu().requires_grad_(True) for k, v in pd.items()}
    to, _ = O.forward(po, tg, O.SegLossConfig())
    assert abs(float(total.detach()) - float(to.detach())) < 1e-5 * abs(float(to.detach()))
    with pytest.raises(RuntimeError):
        crit({"p3": preds["p3"].cuda(), "p4": preds["p4"]}, [t.cuda() for t in tg])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["init", "trained"])
def test_device_kendall_combine(built_lib, tag):
    from mga_yolo_amd import kendall_combine
    z = np.load(os.path.join(GOLDEN, f"kendall_{tag}.npz"))
    det = torch.from_numpy(z["det_loss"]).cuda().requires_grad_(True)
    seg = torch.from_numpy(z["seg_total"]).cuda().requires_grad_(True)
    lv = torch.from_numpy(z["log_vars"]).cuda().requires_grad_(True)
    total = kendall_combine(det, seg, lv)
    assert torch.allclose(total.detach().cpu(), torch.from_numpy(z["total"]), rtol=1e-6, atol=1e-6)
    w = torch.tensor([1.0, -0.5, 2.0]).cuda()
    (total * w).sum().backward()
    e0, e1 = math.exp(-float(z["log_vars"][0])), math.exp(-float(z["log_vars"][1]))
    dl, sg = torch.from_numpy(z["det_loss"]), float(z["seg_total"])
    wc = w.cpu()
    assert torch.allclose(det.grad.cpu(), wc * e0, rtol=1e-6)
    assert abs(float(seg.grad) - float(wc.sum()) * e1) < 1e-5
    want0 = float((wc * (1 - e0 * dl)).sum()); want1 = float(wc.sum()) * (1 - e1 * sg)
    assert abs(float(lv.grad[0]) - want0) < 1e-4 * max(1, abs(want0)) and abs(float(lv.grad[1]) - want1) < 1e-4 * max(1, abs(want1))
    # the recorded gradient (all-ones upstream) as well
    det2, seg2, lv2 = (t.detach().clone().requires_grad_(True) for t in (det, seg, lv))
    kendall_combine(det2, seg2, lv2).sum().backward()
    assert torch.allclose(lv2.grad.cpu(), torch.from_numpy(z["g_log_vars"]), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("ufl", [0, 1])
def test_fused_loss_tail_equals_the_two_calls_bit_for_bit(built_lib, ufl):
    """mgaseg_kendall_forward / _backward (the combine riding in the loss's own launches) against mgaseg_* followed by mgakendall_*,
    through the C-ABI on the same buffers: total, every log entry, dL/dlogits of every level, g_det, g_seg, g_log_vars."""
    import ctypes as C
    from mga_yolo_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    B, sizes = 5, [(24, 20), (12, 10), (6, 5)]
    logits = [torch.randn(B, 1, h, w, generator=g).cuda() for h, w in sizes]
    targets = [(torch.rand(B, 1, 48, 40, generator=g) > 0.7).float().cuda() for _ in sizes]
    det = torch.tensor([1.3, 0.4, 2.2]).cuda(); lv = torch.tensor([0.3, -0.2]).cuda(); g_total = torch.tensor([1.0, -0.5, 2.0]).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def run(fused):
        n = len(sizes)
        levels = (_lib.SegLevel * n)()
        gl = [torch.zeros_like(x) for x in logits]
        for l, (h, w) in enumerate(sizes):
            S = levels[l]
            S.logits, S.target, S.glogits = logits[l].data_ptr(), targets[l].data_ptr(), gl[l].data_ptr()
            S.B, S.H, S.W, S.Ht, S.Wt, S.dtype, S.scale_weight, S.resize = B, h, w, 48, 40, _lib.F32, (1.0, 0.5, 0.25)[l], _lib.SEG_NEAREST
        cfg = _lib.SegCfg(1.0, 1.0, 1.0, 0.7, ufl, 0.5, 0.6, 0.5)
        ws = torch.zeros(lib.mgaseg_ws_bytes(levels, n), dtype=torch.uint8, device="cuda")
        out = torch.zeros(1 + 3 * n, device="cuda"); total = torch.zeros(3, device="cuda")
        g_det = torch.zeros(3, device="cuda"); g_seg = torch.zeros((), device="cuda"); g_lv = torch.zeros(2, device="cuda")
        if fused:
            _lib.check(lib.mgaseg_kendall_forward(levels, n, C.byref(cfg), ws.data_ptr(), ws.numel(), out.data_ptr(), det.data_ptr(), 3, lv.data_ptr(),
                                                  total.data_ptr(), st), "fwd")
            _lib.check(lib.mgaseg_kendall_backward(levels, n, C.byref(cfg), ws.data_ptr(), ws.numel(), out.data_ptr(), det.data_ptr(), 3, lv.data_ptr(),
                                                   g_total.data_ptr(), g_det.data_ptr(), g_seg.data_ptr(), g_lv.data_ptr(), st), "bwd")
        else:
            _lib.check(lib.mgaseg_forward(levels, n, C.byref(cfg), ws.data_ptr(), ws.numel(), out.data_ptr(), st), "fwd")
            _lib.check(lib.mgakendall_forward(det.data_ptr(), 3, out.data_ptr(), lv.data_ptr(), total.data_ptr(), st), "kfwd")
            _lib.check(lib.mgakendall_backward(det.data_ptr(), 3, out.data_ptr(), lv.data_ptr(), g_total.data_ptr(), g_det.data_ptr(),
                                               g_seg.data_ptr(), g_lv.data_ptr(), st), "kbwd")
            _lib.check(lib.mgaseg_backward(levels, n, C.byref(cfg), ws.data_ptr(), ws.numel(), g_seg.data_ptr(), st), "bwd")
        torch.cuda.synchronize()
        return [out, total, g_det, g_seg.reshape(1), g_lv] + gl

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)
    # NULL g_seg is allowed in the fused backward; a NULL total is not
    assert lib.mgaseg_kendall_forward(None, 3, None, None, 0, None, det.data_ptr(), 3, lv.data_ptr(), None, st) != 0
