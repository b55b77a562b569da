"""The layer-loop slice around the hot path, as MGAModel runs it (mga_yolo/model/model.py:57-74, 196-214): per level a mask head
(1x1 conv -> BN -> SiLU -> 3x3 conv, mga_yolo/nn/modules/segmentation.py:81-110) produces mask logits, `[feat, mask]` goes into
MaskCBAM, the logits also go into the multi-scale segmentation loss.  On the device ALL THREE are the HIP paths (this package's
MGAMaskHead, MaskCBAM, SegmentationLoss); the same slice on the host (the modules' host paths + the loss oracle) is the reference.
Checks the seams: dL/dmask from the block flows into the head together with the loss gradient, one backward over all of it."""
import copy

import pytest
import torch

from oracle import segloss_oracle as SO

pytestmark = pytest.mark.gpu


def _slice(feats, heads, blocks, criterion, targets):
    preds, ys = {}, []
    for k, f, h, blk in zip(("p3", "p4", "p5"), feats, heads, blocks):
        m = h(f)                              # mask logits of this level
        preds[k] = m
        ys.append(blk([f, m]))                # the reference's list input
    seg, logs = criterion(preds, targets)
    det_standin = sum((y * y).mean() for y in ys)       # any differentiable consumer of the refined features
    return det_standin + seg, logs, ys


def test_head_block_loss_slice_matches_the_host_statement(built_lib):
    from mga_yolo_amd import MGAMaskHead as _Head, MaskCBAM, SegLossConfig, SegmentationLoss
    torch.manual_seed(0)
    B, lv = 4, [(64, 16, 24, 24), (128, 32, 12, 12), (256, 64, 6, 6)]
    heads = [_Head(c, h) for c, h, _, _ in lv]
    blocks = [MaskCBAM(c) for c, *_ in lv]
    for blk in blocks:
        with torch.no_grad():
            blk.beta.fill_(0.3)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(B, c, H, W, generator=g) for c, _, H, W in lv]
    targets = [(torch.rand(B, 1, H, W, generator=g) > 0.8).float() for _, _, H, W in lv]
    cfg = dict(scale_weights=(1.0, 0.5, 0.25), loss_lambda=0.7)

    # host statement
    h_heads, h_blocks = copy.deepcopy(heads), copy.deepcopy(blocks)
    h_feats = [f.clone().requires_grad_(True) for f in feats]
    crit_h = lambda p, t: SO.forward(p, t, SO.SegLossConfig(**cfg))
    loss_h, logs_h, ys_h = _slice(h_feats, h_heads, h_blocks, crit_h, targets)
    loss_h.backward()

    # device: HIP mask head, HIP block, HIP loss
    d_heads = [copy.deepcopy(h).cuda() for h in heads]
    d_blocks = [copy.deepcopy(b).cuda() for b in blocks]
    d_feats = [f.cuda().requires_grad_(True) for f in feats]
    assert all(h.hip_path() and h.training for h in d_heads)
    loss_d, logs_d, ys_d = _slice(d_feats, d_heads, d_blocks, SegmentationLoss(SegLossConfig(**cfg)), [t.cuda() for t in targets])
    loss_d.backward()

    rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30))
    assert abs(float(loss_d.detach()) - float(loss_h.detach())) < 1e-4 * abs(float(loss_h.detach()))
    for k in logs_h:
        assert abs(logs_d[k] - logs_h[k]) < 1e-4 * max(1.0, abs(logs_h[k])), k
    for l in range(3):
        assert rel(ys_d[l], ys_h[l]) < 1e-4, l
        assert rel(d_feats[l].grad, h_feats[l].grad) < 2e-4, l
        for (n, pd), (_, ph) in zip(d_heads[l].named_parameters(), h_heads[l].named_parameters()):
            assert rel(pd.grad, ph.grad) < 5e-4, (l, n)          # carries dL/dmask of the block + the loss gradient, through BN
        for (n, pd), (_, ph) in zip(d_blocks[l].named_parameters(), h_blocks[l].named_parameters()):
            assert rel(pd.grad, ph.grad) < 5e-4, (l, n)


@pytest.mark.parametrize("amp_dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_layer_loop_slice_under_autocast_with_grad_scaler(built_lib, amp_dtype, tol):
    """The reference trainer's DEFAULT numeric mode (U/engine/trainer.py:355-365 `with autocast(self.amp)`, :477-488
    `self.scaler.scale(self.loss).backward()`): fp32 parameters, the layer loop under torch.autocast, half-precision features and mask
    logits, gradients through a GradScaler.  Device: MGAMaskHead -> [feat, logits] -> MaskCBAM -> SegmentationLoss, all HIP paths.
    Checked against the fp32 host statement evaluated on the SAME half-rounded tensors (features, logits, refined features), at the
    tolerances of the half-precision I/O tests; the output dtypes must be what the reference's ops produce under autocast (half y, half
    logits, fp32 loss, fp32 parameter gradients)."""
    from mga_yolo_amd import MGAMaskHead as _Head, MaskCBAM, SegLossConfig, SegmentationLoss
    torch.manual_seed(1)
    B, lv = 4, [(64, 16, 40, 40), (128, 32, 20, 20), (256, 64, 10, 10)]
    heads = [_Head(c, h) for c, h, _, _ in lv]
    blocks = [MaskCBAM(c) for c, *_ in lv]
    for blk in blocks:
        with torch.no_grad():
            blk.beta.fill_(0.3)
    g = torch.Generator().manual_seed(6)
    feats = [torch.randn(B, c, H, W, generator=g).to(amp_dtype).float() for c, _, H, W in lv]      # what a half-precision backbone hands over
    gys = [torch.randn(B, c, H, W, generator=g).to(amp_dtype).float() / (c * H * W) for c, _, H, W in lv]
    targets = [(torch.rand(B, 1, H, W, generator=g) > 0.8).float() for _, _, H, W in lv]
    cfg = dict(scale_weights=(1.0, 0.5, 0.25), loss_lambda=0.7)
    rnd = lambda t: t + (t.to(amp_dtype).float() - t).detach()      # the device's rounding points, straight-through for the gradient

    # fp32 host statement on the rounded tensors
    h_heads, h_blocks = copy.deepcopy(heads), copy.deepcopy(blocks)
    h_feats = [f.clone().requires_grad_(True) for f in feats]
    preds, ys_h, logits_h = {}, [], []
    for k, f, h, blk in zip(("p3", "p4", "p5"), h_feats, h_heads, h_blocks):
        m = rnd(h(f))
        preds[k] = m
        logits_h.append(m)
        ys_h.append(rnd(blk([f, m])))
    seg_h, logs_h = SO.forward(preds, targets, SO.SegLossConfig(**cfg))
    loss_h = sum((y * gy).sum() for y, gy in zip(ys_h, gys)) + seg_h
    loss_h.backward()

    # device: fp32 parameters, autocast, GradScaler -- what an unchanged MGATrainer does on a GPU
    d_heads = [copy.deepcopy(h).cuda() for h in heads]
    d_blocks = [copy.deepcopy(b).cuda() for b in blocks]
    leaves = [f.cuda().requires_grad_(True) for f in feats]
    params = [p for mod in d_heads + d_blocks for p in mod.parameters()]
    opt = torch.optim.SGD(params, lr=0.0)
    scaler = torch.amp.GradScaler("cuda", init_scale=256.0)
    crit = SegmentationLoss(SegLossConfig(**cfg))
    with torch.autocast("cuda", dtype=amp_dtype):
        preds_d, ys_d, logits_d = {}, [], []
        for k, leaf, h, blk in zip(("p3", "p4", "p5"), leaves, d_heads, d_blocks):
            f = leaf.to(amp_dtype)                                    # the half-precision feature map of the neck
            m = h(f)
            preds_d[k] = m
            logits_d.append(m)
            ys_d.append(blk([f, m]))
        seg_d, logs_d = crit(preds_d, [t.cuda() for t in targets])
        loss_d = sum((y.float() * gy.cuda()).sum() for y, gy in zip(ys_d, gys)) + seg_d
    assert all(y.dtype == amp_dtype for y in ys_d) and all(m.dtype == amp_dtype for m in logits_d)
    assert seg_d.dtype == torch.float32 and loss_d.dtype == torch.float32
    scaler.scale(loss_d).backward()
    scale = scaler.get_scale()
    scaler.unscale_(opt)
    assert all(p.grad is not None and p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()) for p in params)
    scaler.step(opt)
    scaler.update()
    assert scaler.get_scale() == scale                                # no inf / nan was found: the step was not skipped

    rel = lambda a, b: float((a.detach().cpu().double() - b.detach().double()).abs().max() / b.detach().double().abs().max().clamp_min(1e-30))
    errs = {}
    errs["seg"] = abs(float(seg_d) - float(seg_h)) / abs(float(seg_h))
    for l in range(3):
        errs[f"logits{l}"] = rel(logits_d[l].float(), logits_h[l])
        errs[f"y{l}"] = rel(ys_d[l].float(), ys_h[l])
        errs[f"gfeat{l}"] = rel(leaves[l].grad / scale, h_feats[l].grad)
        for (n, pd), (_, ph) in zip(list(d_heads[l].named_parameters()) + list(d_blocks[l].named_parameters()),
                                    list(h_heads[l].named_parameters()) + list(h_blocks[l].named_parameters())):
            errs[f"g{l}.{n}"] = rel(pd.grad, ph.grad)
    print("amp errors", amp_dtype, {k: f"{v:.2e}" for k, v in errs.items()})
    bad = {k: v for k, v in errs.items() if not v < (tol if k.startswith(("logits", "y", "seg")) else 4 * tol)}
    assert not bad, bad
    # the BatchNorm running statistics were updated once, in fp32, from the half features
    for l in range(3):
        assert rel(d_heads[l].proj[1].running_mean, h_heads[l].proj[1].running_mean) < tol
        assert int(d_heads[l].proj[1].num_batches_tracked) == 1
