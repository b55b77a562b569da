"""CPU oracle for MaskECA  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as maskcbam_oracle.py).

Restates ``mga_yolo/nn/modules/masked_eca.py``:
  * ``eca_kernel_size`` :44-54   adaptive odd 1-D kernel size from the channel count
  * ``_pool``           :139-165 masked average pooling with GAP fallback for tiny masks (``.any()`` branch :152-161 and the
                                 plain branch :163-165 are the same arithmetic whenever every sample is valid)
  * ``forward``         :167-196 conv1d over the channel axis -> sigmoid -> g = 1 + softplus(beta) * (w - 0.5) -> x * g
and the gradients autograd derives for them.  Pinned by ``tests/golden/eca_*.npz`` (outputs of the reference module itself,
``oracle/gen_golden.py``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn.functional as F


def eca_kernel_size(channels: int, gamma: float = 2.0, b: float = 1.0, k_min: int = 3, k_max: int = 15) -> int:
    """masked_eca.py:44-54"""
    if channels <= 0:
        return k_min
    k = int(abs((channels.bit_length() - 1) / gamma + b))
    k = max(k_min, min(k, k_max))
    return k if k % 2 == 1 else k + 1


@dataclass
class EcaParams:
    w: torch.Tensor      # conv1d.weight (1, 1, k)
    beta: torch.Tensor   # beta ()

    @staticmethod
    def default_init(channels: int, seed: int = 0) -> "EcaParams":
        import torch.nn as nn
        torch.manual_seed(seed)
        k = eca_kernel_size(channels)
        cv = nn.Conv1d(1, 1, kernel_size=k, padding=k // 2, bias=False)
        return EcaParams(cv.weight.detach().clone(), torch.zeros((), dtype=torch.float32))


@dataclass
class EcaConfig:
    use_sigmoid_mask: bool = True
    tiny_thr: float = 1e-4
    eps: float = 1e-6


def forward(x: torch.Tensor, mask: Optional[torch.Tensor], p: EcaParams, cfg: EcaConfig = EcaConfig()):
    B, C, H, W = x.shape
    N = H * W
    dt = x.dtype
    xf = x.reshape(B, C, N)
    gap = xf.mean(dim=2)
    t = {}
    if mask is None:
        avg, s = gap, None
        use = torch.zeros(B, dtype=dt); den = torch.ones(B, dtype=dt); S = torch.zeros(B, dtype=dt); mavg = gap
    else:
        m = mask.unsqueeze(1) if mask.dim() == 3 else mask
        if tuple(m.shape) != (B, 1, H, W):
            raise RuntimeError(f"mask shape {tuple(mask.shape)} does not match feature")
        m2 = m.reshape(B, N).to(dt)
        s = torch.sigmoid(m2) if cfg.use_sigmoid_mask else m2                       # masked_eca.py:146-147
        S = s.sum(dim=1)
        use = ((S / N) >= cfg.tiny_thr).to(dt)                                      # :152-153, 159
        den = S.clamp_min(cfg.eps)                                                  # :156, 164
        mavg = (xf * s[:, None, :]).sum(dim=2) / den[:, None]                       # :157, 165
        avg = mavg * use[:, None] + gap * (1.0 - use[:, None])                      # :161
    k = p.w.shape[-1]
    y1 = F.conv1d(avg.unsqueeze(1), p.w.to(dt), padding=k // 2).squeeze(1)          # :181-185
    w = torch.sigmoid(y1)                                                           # :187
    a = F.softplus(p.beta.to(dt))
    g = 1.0 + a * (w - 0.5)                                                         # :190
    y = xf * g[:, :, None]                                                          # :193
    t.update(S=S, use=use, den=den, mavg=mavg, avg=avg, w=w, g=g, a=a, s=s)
    return y.reshape(B, C, H, W), t


def backward(gy: torch.Tensor, x: torch.Tensor, mask: Optional[torch.Tensor], p: EcaParams, cfg: EcaConfig, t: dict):
    B, C, H, W = x.shape
    N = H * W
    dt = x.dtype
    xf, g_ = x.reshape(B, C, N), gy.reshape(B, C, N)
    w, g, a, avg = t["w"], t["g"], t["a"], t["avg"]
    gg = (g_ * xf).sum(dim=2)                                   # dL/dg  (B,C)
    gbeta = torch.sigmoid(p.beta.to(dt)) * (gg * (w - 0.5)).sum()
    gy1 = a * gg * w * (1.0 - w)                                # dL/d(conv output)
    k = p.w.shape[-1]
    pad = k // 2
    avgp = F.pad(avg, (pad, pad))
    gw = torch.stack([(gy1 * avgp[:, j:j + C]).sum() for j in range(k)]).reshape(1, 1, k)
    g_avg = F.conv_transpose1d(gy1.unsqueeze(1), p.w.to(dt), padding=pad).squeeze(1)
    gx = g_ * g[:, :, None]
    gmask = None
    if mask is None:
        gx = gx + g_avg[:, :, None] / N
    else:
        s, use, den, S = t["s"], t["use"], t["den"], t["S"]
        gx = gx + g_avg[:, :, None] * ((use / den)[:, None, None] * s[:, None, :] + ((1.0 - use) / N)[:, None, None])
        live = (S >= cfg.eps).to(dt)[:, None]
        gs = (use / den)[:, None] * ((g_avg[:, :, None] * xf).sum(dim=1) - (g_avg * t["mavg"]).sum(dim=1, keepdim=True) * live)
        gmask = (gs * s * (1.0 - s) if cfg.use_sigmoid_mask else gs).reshape(mask.shape)
    return dict(gx=gx.reshape(B, C, H, W), gmask=gmask, gw=gw, gbeta=gbeta)


def reference_form(x, mask, p: EcaParams, cfg: EcaConfig = EcaConfig()):
    """Differentiable eager-op form (one op per reference line); CPU baseline + autograd cross-check."""
    B, C, H, W = x.shape
    if mask is None:
        v = F.adaptive_avg_pool2d(x, 1).view(B, C)
    else:
        m = mask.unsqueeze(1) if mask.dim() == 3 else mask
        s = m.sigmoid() if cfg.use_sigmoid_mask else m
        se = s.expand(B, C, H, W)
        valid = (se.mean(dim=(2, 3)).mean(dim=1) >= cfg.tiny_thr).to(x.dtype).unsqueeze(1)
        masked = (x * se).sum(dim=(2, 3)) / se.sum(dim=(2, 3)).clamp_min(cfg.eps)
        v = masked * valid + F.adaptive_avg_pool2d(x, 1).view(B, C) * (1.0 - valid)
    k = p.w.shape[-1]
    w = F.conv1d(v.unsqueeze(1), p.w, padding=k // 2).squeeze(1).sigmoid().view(B, C, 1, 1)
    return x * (1.0 + F.softplus(p.beta) * (w - 0.5))


def reference_form_step(x, mask, p: EcaParams, cfg: EcaConfig, gy):
    leaves = [p.w.detach().clone().requires_grad_(True), p.beta.detach().clone().requires_grad_(True)]
    xr = x.detach().clone().requires_grad_(True)
    mr = None if mask is None else mask.detach().clone().requires_grad_(True)
    y = reference_form(xr, mr, EcaParams(*leaves), cfg)
    y.backward(gy)
    return y.detach(), dict(gx=xr.grad, gmask=None if mr is None else mr.grad, gw=leaves[0].grad, gbeta=leaves[1].grad)
