"""TEST INFRASTRUCTURE — CPU restatement of the reference's multi-scale segmentation loss, default mode.

Follows mga_yolo/nn/losses/segmentation.py: `_dice_probs` :38-42, `forward` :87-151 (BCEWithLogitsLoss(reduction="mean") :36,
soft Dice on sigmoid(pred) :135, scale weights :112, nearest target resize :103-110, loss_lambda :149).  Only tests/,
__graft_entry__.smoke() and the benchmarks' CPU-baseline leg may import this.

PINNED by outputs of the reference itself: ``oracle/gen_golden_segloss.py`` imported the reference's own ``SegmentationLoss`` in the
build container (its module-scope LOGGER import needs cv2, absent here; the in-process stand-in of SURVEY appendix A2 satisfies it and
is never executed) and stored logits, targets, total, every log entry and d total / d logits for 16 cases -- both modes, nearest and
bilinear (MGA_PROB_MODE) target resize, 3-D targets, missing levels, weights -- in ``tests/golden/segloss_*.npz``; the Kendall combine
of ``MGAModel.loss`` (model/model.py:204-206), evaluated by the reference model, in ``tests/golden/kendall_*.npz``.
``tests/test_segloss.py`` checks this restatement against them.  Unified-Focal mode (`_lmf` :44-63, `_lmft` :65-85) is restated too.
"""
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class SegLossConfig:                       # segmentation.py:9-21
    bce_weight: float = 1.0
    dice_weight: float = 1.0
    scale_weights: Sequence[float] = (1.0, 1.0, 1.0)
    smooth: float = 1.0
    loss_lambda: float = 1.0
    enabled: bool = True
    use_unified_focal: bool = False
    ufl_lambda: float = 0.5
    ufl_delta: float = 0.6
    ufl_gamma: float = 0.5


def dice_probs(probs, tgt, smooth):        # segmentation.py:38-42
    inter = (probs * tgt).sum(dim=(1, 2, 3))
    denom = probs.sum(dim=(1, 2, 3)) + tgt.sum(dim=(1, 2, 3)) + smooth
    return 1.0 - (2.0 * inter + smooth) / denom


def lmf(logits, tgt, delta, gamma, eps=1e-6):          # segmentation.py:44-63
    x, t = logits.float(), tgt.float()
    probs = torch.sigmoid(x)
    pt = torch.where(t > 0.5, probs, 1.0 - probs).clamp(eps, 1.0 - eps)
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none").float()
    w = torch.where(t > 0.5, delta, 1.0 - delta).float()
    base = (1.0 - pt).clamp_min(eps)
    return (base.pow(1.0 - gamma) * ce * w).mean()


def lmft(logits, tgt, delta, gamma, smooth, eps=1e-6):  # segmentation.py:65-85
    x, t = logits.float(), tgt.float()
    p = torch.sigmoid(x)
    tp = (p * t).sum(dim=(1, 2, 3))
    fn = (t * (1.0 - p)).sum(dim=(1, 2, 3))
    fp = ((1.0 - t) * p).sum(dim=(1, 2, 3))
    denom = (tp + delta * fn + (1.0 - delta) * fp + smooth).clamp_min(eps)
    mti = (tp + smooth) / denom
    return (1.0 - mti).clamp_min(eps).pow(gamma).mean()


def forward(preds: Dict[str, torch.Tensor], targets: List[torch.Tensor], cfg: SegLossConfig, bilinear_targets: bool = False
            ) -> Tuple[torch.Tensor, Dict[str, float]]:
    """segmentation.py:87-151.  `bilinear_targets` stands for the MGA_PROB_MODE environment switch (:103-108)."""
    first = next(iter(preds.values()))
    if not cfg.enabled:
        return torch.zeros((), device=first.device), {}
    total = torch.zeros((), device=first.device, dtype=torch.float32)
    logs: Dict[str, float] = {}
    for i, sk in enumerate(["p3", "p4", "p5"]):
        if sk not in preds or i >= len(targets):
            continue
        pred, tgt = preds[sk], targets[i]
        if tgt.dim() == 3:
            tgt = tgt.unsqueeze(1)
        if tgt.shape[-2:] != pred.shape[-2:]:
            if bilinear_targets:
                tgt = F.interpolate(tgt.float(), size=pred.shape[-2:], mode="bilinear", align_corners=False)
            else:
                tgt = F.interpolate(tgt.float(), size=pred.shape[-2:], mode="nearest")
        w_scale = cfg.scale_weights[i] if i < len(cfg.scale_weights) else 1.0
        if cfg.use_unified_focal:
            a = lmf(pred.float(), tgt.float(), cfg.ufl_delta, cfg.ufl_gamma)
            b = lmft(pred.float(), tgt.float(), cfg.ufl_delta, cfg.ufl_gamma, cfg.smooth)
            combined = w_scale * (cfg.ufl_lambda * a + (1.0 - cfg.ufl_lambda) * b)
        else:
            a = F.binary_cross_entropy_with_logits(pred, tgt.float(), reduction="mean")
            b = dice_probs(torch.sigmoid(pred), tgt.float(), cfg.smooth).mean()
            combined = w_scale * (cfg.bce_weight * a + cfg.dice_weight * b)
        logs[f"{sk}_bce"], logs[f"{sk}_dice"] = float(a.detach()), float(b.detach())
        if not torch.isfinite(combined):
            raise FloatingPointError("Segmentation loss became non-finite.")
        total = total + combined.float()
        logs[f"{sk}_combined"] = float(combined.detach())
    total = total * cfg.loss_lambda
    logs["seg_total"] = float(total.detach())
    return total, logs


def kendall_combine(det_loss: torch.Tensor, seg_total: torch.Tensor, log_vars: torch.Tensor) -> torch.Tensor:
    """mga_yolo/model/model.py:204-206: L = e^{-s_det} L_det + s_det + e^{-s_seg} L_seg + s_seg (det_loss is the criterion's vector)."""
    s_det, s_seg = log_vars[0], log_vars[1]
    return torch.exp(-s_det) * det_loss + s_det + torch.exp(-s_seg) * seg_total + s_seg
