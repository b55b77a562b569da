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
