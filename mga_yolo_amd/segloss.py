"""Multi-scale segmentation loss of MGA-YOLO behind the reference's interface.

Mirrors mga_yolo/nn/losses/segmentation.py: `SegLossConfig` :9-21 and `SegmentationLoss` :23-151 (same constructor, same
`forward(preds: {"p3"|"p4"|"p5": logits}, targets: [mask, ...]) -> (total, logs)`, same log keys, same FloatingPointError on a
non-finite value).  Device tensors (both modes: BCE-with-logits + soft Dice, and Unified Focal) go through the HIP entry points
`mgaseg_forward` / `mgaseg_backward` (include/mgacbam.h): two launches forward and one backward for all levels, the nearest target
resize done inside the kernels with F.interpolate's index rule.  Host tensors use the same torch ops as the reference.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

SCALE_KEYS = ("p3", "p4", "p5")
_DTYPE_CODES = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


@dataclass
class SegLossConfig:
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


class _SegFn(torch.autograd.Function):
    """flat = n x (logits, target); returns out = [total, (bce, dice, combined) per level] on the device."""

    @staticmethod
    def forward(ctx, cfg: tuple, weights: Tuple[float, ...], resize: Tuple[int, ...], *flat):
        n = len(flat) // 2
        lib = _lib.load()
        levels = (_lib.SegLevel * n)()
        keep = []
        dev = flat[0].device
        for l in range(n):
            x, t = flat[2 * l], flat[2 * l + 1]
            xc = x.contiguous()
            tc = t.detach().to(torch.float32).contiguous()
            B, _, H, W = xc.shape
            L = levels[l]
            L.logits, L.target, L.glogits = xc.data_ptr(), tc.data_ptr(), None
            L.B, L.H, L.W, L.Ht, L.Wt = B, H, W, tc.shape[-2], tc.shape[-1]
            L.dtype, L.scale_weight, L.resize = _DTYPE_CODES[xc.dtype], weights[l], resize[l]
            keep += [xc, tc]
        c = _lib.SegCfg(*cfg)
        ws = torch.empty(lib.mgaseg_ws_bytes(levels, n), dtype=torch.uint8, device=dev)
        out = torch.empty(1 + 3 * n, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.mgaseg_forward(levels, n, C.byref(c), ws.data_ptr(), ws.numel(), out.data_ptr(),
                                          torch.cuda.current_stream(dev).cuda_stream), "mgaseg_forward")
        ctx.save_for_backward(ws, *keep)
        ctx.cfg, ctx.weights, ctx.resize = cfg, weights, resize
        return out

    @staticmethod
    def backward(ctx, gout):
        ws, *keep = ctx.saved_tensors
        n = len(keep) // 2
        lib = _lib.load()
        levels = (_lib.SegLevel * n)()
        dev = ws.device
        grads = []
        for l in range(n):
            xc, tc = keep[2 * l], keep[2 * l + 1]
            gx = torch.empty_like(xc)
            B, _, H, W = xc.shape
            L = levels[l]
            L.logits, L.target, L.glogits = xc.data_ptr(), tc.data_ptr(), gx.data_ptr()
            L.B, L.H, L.W, L.Ht, L.Wt = B, H, W, tc.shape[-2], tc.shape[-1]
            L.dtype, L.scale_weight, L.resize = _DTYPE_CODES[xc.dtype], ctx.weights[l], ctx.resize[l]
            grads += [gx, None]
        g0 = gout[0:1].to(torch.float32).contiguous()       # only `total` is differentiable; the log entries are detached copies
        c = _lib.SegCfg(*ctx.cfg)
        with torch.cuda.device(dev):
            _lib.check(lib.mgaseg_backward(levels, n, C.byref(c), ws.data_ptr(), ws.numel(), g0.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream), "mgaseg_backward")
        return (None, None, None, *grads)


def _dice_probs(probs, tgt, smooth):
    inter = (probs * tgt).sum(dim=(1, 2, 3))
    denom = probs.sum(dim=(1, 2, 3)) + tgt.sum(dim=(1, 2, 3)) + smooth
    return 1.0 - (2.0 * inter + smooth) / denom


def _lmf(x, t, delta, gamma, eps=1e-6):
    probs = torch.sigmoid(x)
    pt = torch.where(t > 0.5, probs, 1.0 - probs).clamp(eps, 1.0 - eps)
    ce = F.binary_cross_entropy_with_logits(x, t, reduction="none").float()
    w = torch.where(t > 0.5, delta, 1.0 - delta).float()
    return ((1.0 - pt).clamp_min(eps).pow(1.0 - gamma) * ce * w).mean()


def _lmft(x, t, delta, gamma, smooth, eps=1e-6):
    p = torch.sigmoid(x)
    tp = (p * t).sum(dim=(1, 2, 3))
    fn = (t * (1.0 - p)).sum(dim=(1, 2, 3))
    fp = ((1.0 - t) * p).sum(dim=(1, 2, 3))
    mti = (tp + smooth) / (tp + delta * fn + (1.0 - delta) * fp + smooth).clamp_min(eps)
    return (1.0 - mti).clamp_min(eps).pow(gamma).mean()


class SegmentationLoss(nn.Module):
    def __init__(self, cfg: SegLossConfig) -> None:
        super().__init__()
        self.cfg = cfg

    def forward(self, preds: Dict[str, torch.Tensor], targets: List[torch.Tensor]) -> Tuple[torch.Tensor, Dict[str, float]]:
        cfg = self.cfg
        first = next(iter(preds.values()))
        if not cfg.enabled:
            return torch.zeros((), device=first.device), {}
        used = []
        prob_mode = bool(os.getenv("MGA_PROB_MODE", False))
        for i, sk in enumerate(SCALE_KEYS):
            if sk not in preds or i >= len(targets):
                continue
            pred, tgt = preds[sk], targets[i]
            if tgt.dim() == 3:
                tgt = tgt.unsqueeze(1)
            w = cfg.scale_weights[i] if i < len(cfg.scale_weights) else 1.0
            used.append((sk, pred, tgt, float(w)))
        if not used:
            return torch.zeros((), device=first.device, dtype=torch.float32) * cfg.loss_lambda, {"seg_total": 0.0}
        on_device = [p.is_cuda for _, p, _, _ in used]
        if all(on_device):
            # device tensors have no other path.  Levels of different element types (never produced by the reference's layer loop,
            # which runs all three mask heads under one autocast state) are computed in fp32, like the loss's own accumulators
            if len({p.dtype for _, p, _, _ in used}) != 1 or used[0][1].dtype not in _DTYPE_CODES:
                used = [(sk, p.float(), t, w) for sk, p, t, w in used]
            return self._device_forward(used, prob_mode)
        if any(on_device):
            raise RuntimeError("SegmentationLoss: logits of one call must all live on the GPU or all on the host")
        return self._torch_forward(used, prob_mode)

    # ---- HIP path ---------------------------------------------------------------------------------------------------------
    def _device_forward(self, used, prob_mode: bool):
        cfg = self.cfg
        flat = []
        # targets at another resolution are gathered inside the kernels: nearest, or bilinear for probabilistic masks (segmentation.py:103-110)
        mode = _lib.SEG_BILINEAR if prob_mode else _lib.SEG_NEAREST
        for _, pred, tgt, _ in used:
            if pred.dim() != 4 or pred.shape[1] != 1 or tgt.shape[0] != pred.shape[0] or tgt.shape[1] != 1:
                raise RuntimeError(f"SegmentationLoss: logits {tuple(pred.shape)} / target {tuple(tgt.shape)} must be (B,1,H,W)")
            flat += [pred, tgt.to(pred.device)]
        out = _SegFn.apply((float(cfg.bce_weight), float(cfg.dice_weight), float(cfg.smooth), float(cfg.loss_lambda),
                            int(bool(cfg.use_unified_focal)), float(cfg.ufl_lambda), float(cfg.ufl_delta), float(cfg.ufl_gamma)),
                           tuple(w for *_, w in used), tuple(mode for _ in used), *flat)
        vals = out.detach().cpu().tolist()                          # ONE device->host copy for every log entry
        logs: Dict[str, float] = {}
        for l, (sk, *_rest) in enumerate(used):
            bce, dice, comb = vals[1 + 3 * l: 4 + 3 * l]
            if not (comb == comb and abs(comb) != float("inf")):
                raise FloatingPointError("Segmentation loss became non-finite.")
            logs[f"{sk}_bce"], logs[f"{sk}_dice"], logs[f"{sk}_combined"] = bce, dice, comb
        logs["seg_total"] = vals[0]
        return out[0], logs

    # ---- torch ops (host tensors) --------------------------------------------------------------------------
    def _torch_forward(self, used, prob_mode: bool = False):
        cfg = self.cfg
        total = torch.zeros((), device=used[0][1].device, dtype=torch.float32)
        logs: Dict[str, float] = {}
        for sk, pred, tgt, w in used:
            if tgt.shape[-2:] != pred.shape[-2:]:
                if prob_mode:                                       # probabilistic masks: bilinear, as the reference (:103-108)
                    tgt = F.interpolate(tgt.float(), size=pred.shape[-2:], mode="bilinear", align_corners=False)
                else:
                    tgt = F.interpolate(tgt.float(), size=pred.shape[-2:], mode="nearest")
            if cfg.use_unified_focal:
                a = _lmf(pred.float(), tgt.float(), cfg.ufl_delta, cfg.ufl_gamma)
                b = _lmft(pred.float(), tgt.float(), cfg.ufl_delta, cfg.ufl_gamma, cfg.smooth)
                comb = w * (cfg.ufl_lambda * a + (1.0 - cfg.ufl_lambda) * b)
            else:
                a = F.binary_cross_entropy_with_logits(pred, tgt.float(), reduction="mean")
                b = _dice_probs(torch.sigmoid(pred), tgt.float(), cfg.smooth).mean()
                comb = w * (cfg.bce_weight * a + cfg.dice_weight * b)
            logs[f"{sk}_bce"], logs[f"{sk}_dice"] = float(a.detach()), float(b.detach())
            if not torch.isfinite(comb):
                raise FloatingPointError("Segmentation loss became non-finite.")
            total = total + comb.float()
            logs[f"{sk}_combined"] = float(comb.detach())
        total = total * cfg.loss_lambda
        logs["seg_total"] = float(total.detach())
        return total, logs


# ---------------------------------------------------------------------------------------------------------------------------
# Kendall multi-task combine of MGAModel.loss (mga_yolo/model/model.py:204-206)
# ---------------------------------------------------------------------------------------------------------------------------
class _KendallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, det, seg, log_vars):
        lib = _lib.load()
        d, s_, lv = det.detach().float().contiguous(), seg.detach().float().reshape(()).contiguous(), log_vars.detach().float().contiguous()
        total = torch.empty_like(d)
        with torch.cuda.device(d.device):
            _lib.check(lib.mgakendall_forward(d.data_ptr(), d.numel(), s_.data_ptr(), lv.data_ptr(), total.data_ptr(),
                                              torch.cuda.current_stream(d.device).cuda_stream), "mgakendall_forward")
        ctx.save_for_backward(d, s_, lv)
        ctx.shapes = (det.shape, seg.shape)
        return total.view(det.shape)

    @staticmethod
    def backward(ctx, g_total):
        d, s_, lv = ctx.saved_tensors
        lib = _lib.load()
        g = g_total.float().contiguous()
        g_det, g_seg, g_lv = torch.empty_like(d), torch.empty_like(s_), torch.empty_like(lv)
        with torch.cuda.device(d.device):
            _lib.check(lib.mgakendall_backward(d.data_ptr(), d.numel(), s_.data_ptr(), lv.data_ptr(), g.data_ptr(), g_det.data_ptr(),
                                               g_seg.data_ptr(), g_lv.data_ptr(), torch.cuda.current_stream(d.device).cuda_stream),
                       "mgakendall_backward")
        return g_det.view(ctx.shapes[0]), g_seg.view(ctx.shapes[1]), g_lv


def kendall_combine(det_loss: torch.Tensor, seg_total: torch.Tensor, log_vars: torch.Tensor) -> torch.Tensor:
    """``exp(-s_det) * det_loss + s_det + exp(-s_seg) * seg_total + s_seg`` with ``log_vars = [s_det, s_seg]`` (the learnable
    ``mtl_log_vars``): the multi-task combine of ``MGAModel.loss`` (mga_yolo/model/model.py:204-206).  ``det_loss`` is the detection
    criterion's vector (box, cls, dfl), the result has its shape.  Device tensors: one launch forward, one backward
    (``mgakendall_*``), gradients to det_loss, seg_total and both log-variances; host tensors: the same three torch ops."""
    if log_vars.numel() != 2:
        raise ValueError("log_vars must hold [s_det, s_seg]")
    if det_loss.is_cuda:
        if not (seg_total.is_cuda and log_vars.is_cuda):
            raise RuntimeError("kendall_combine: det_loss, seg_total and log_vars must live on the same device")
        return _KendallFn.apply(det_loss, seg_total, log_vars)
    s_det, s_seg = log_vars[0], log_vars[1]
    return torch.exp(-s_det) * det_loss + s_det + torch.exp(-s_seg) * seg_total + s_seg
