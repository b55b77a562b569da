"""Host-side mirror of the reference's module interface for the hot path.

``MaskCBAM`` keeps the contract of ``mga_yolo/nn/modules/masked_cbam.py:10-174``: same constructor signature and
defaults, same parameter names / shapes / creation order (so the same seed gives the same initial values and reference
checkpoints load: ``beta``, ``cam_mlp.0.{weight,bias}``, ``cam_mlp.2.{weight,bias}``, ``sam_conv.weight``), ``[feat, mask]``
list input, ``alpha`` property, ``extra_repr``, plain-tensor output (forward hooks work), deepcopy / pickle safe (no ctypes
state on the instance -- EMA deep-copies the module, U/utils/torch_utils.py:722).

Device tensors run the hand-written HIP kernels through ``functional.mask_cbam`` (one library call forward, one backward).
Host (CPU) tensors -- the 256x256 stride probe ``parse_model`` runs at build time (U/nn/tasks.py:413-429) and BASELINE
config 0 (``device='cpu'``) -- run ``_host_forward``, a plain-PyTorch statement of the same mathematics; it is never
used for a device tensor, and a device tensor with the library missing raises.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .functional import BlockConfig, EcaConfig, mask_cbam, mask_eca, mask_head, prob_mask_gate

_GATER_MODES = ("deterministic", "gumbel", "hard_st", "bernoulli_detach")


class ProbMaskGater(nn.Module):
    """Stochastic gate applied to the mask BEFORE the block when ``MGA_PROB_MODE`` is set
    (reference mga_yolo/nn/modules/probmaskgater.py:8-98): the input is clamped to [0,1] (note: applied to whatever the mask
    head emits, logits included), eval / 'deterministic' return that, 'gumbel' adds logistic noise in logit space, 'hard_st'
    thresholds with a straight-through gradient, 'bernoulli_detach' samples without gradient.  The random numbers always come
    from torch's generator, drawn exactly as the reference draws them; for device tensors in the two Gumbel modes everything
    after the draws is ONE HIP launch (``functional.prob_mask_gate``) instead of ~14 elementwise kernels."""

    def __init__(self, mode: str = "gumbel", tau: float = 1.0, p_min: float = 0.0, threshold: float = 0.5,
                 seed: Optional[int] = None):
        super().__init__()
        if tau <= 0:
            raise ValueError("tau must be > 0")
        self.mode, self.tau, self.p_min, self.threshold, self._seed = mode, float(tau), float(p_min), float(threshold), seed
        if seed is not None:
            self.register_buffer("_ctr", torch.zeros((), dtype=torch.long))

    def _generator(self, device):
        if self._seed is None:
            return None
        g = torch.Generator(device=device)
        g.manual_seed(self._seed + int(self._ctr.item()))
        self._ctr.add_(1)
        return g

    def _soft_sample(self, p: torch.Tensor) -> torch.Tensor:
        gen = self._generator(p.device)
        lo, hi = 1e-6, 1.0 - 1e-6
        u1 = torch.rand(p.shape, dtype=p.dtype, device=p.device, generator=gen).clamp_(lo, hi)
        u2 = torch.rand(p.shape, dtype=p.dtype, device=p.device, generator=gen).clamp_(lo, hi)
        noise = torch.log(-torch.log(u2)) - torch.log(-torch.log(u1))          # Gumbel(u1) - Gumbel(u2): logistic
        q = p.clamp(lo, hi)
        logit = torch.log(q) - torch.log1p(-q)
        return torch.sigmoid((logit + noise) / self.tau)

    def forward(self, p: torch.Tensor) -> torch.Tensor:
        if p.dim() == 3:
            p = p.unsqueeze(1)
        if p.is_cuda and self.training and self.mode in ("gumbel", "hard_st"):
            p = p.float()
            gen = self._generator(p.device)
            u1 = torch.rand(p.shape, dtype=p.dtype, device=p.device, generator=gen)
            u2 = torch.rand(p.shape, dtype=p.dtype, device=p.device, generator=gen)
            return prob_mask_gate(p, u1, u2, self.tau, self.p_min, self.threshold, hard=self.mode == "hard_st")
        p = p.float().clamp(0.0, 1.0)
        if self.p_min > 0:
            p = p.clamp_min(self.p_min)
        if not self.training or self.mode == "deterministic":
            return p
        if self.mode == "gumbel":
            return self._soft_sample(p)
        if self.mode == "hard_st":
            soft = self._soft_sample(p)
            return (soft > self.threshold).float() + (soft - soft.detach())
        if self.mode == "bernoulli_detach":
            return torch.bernoulli(p.detach(), generator=self._generator(p.device))
        return p


class MaskCBAM(nn.Module):
    """Mask-guided CBAM at one pyramid level: masked avg/max channel attention -> spatial attention that also sees the
    mask -> ``x + softplus(beta) * (refined - x)``.  Drop-in for the reference class of the same name."""

    def __init__(self, channels: int, r: int = 16, spatial_k: int = 7, use_sigmoid_mask: bool = True,
                 tiny_mask_thr: float = 1e-4, eps: float = 1e-6) -> None:
        super().__init__()
        assert r > 0 and channels > 0
        self.C = channels
        self.r = r
        self.k = spatial_k if spatial_k % 2 == 1 else spatial_k + 1
        self.use_sigmoid_mask = use_sigmoid_mask
        self.tiny_thr = tiny_mask_thr
        self.eps = eps
        hidden = max(1, channels // r)
        # containers only hold the parameters (same names/order/init as the reference); the math runs in the kernels
        self.cam_mlp = nn.Sequential(nn.Linear(channels, hidden, bias=True), nn.ReLU(inplace=True),
                                     nn.Linear(hidden, channels, bias=True))
        self.sam_conv = nn.Conv2d(3, 1, kernel_size=self.k, padding=self.k // 2, bias=False)
        self.beta = nn.Parameter(torch.zeros((), dtype=torch.float32))
        if os.getenv("MGA_PROB_MODE", False):                                   # masked_cbam.py:67-78 (truthy for any string)
            approach = os.getenv("MGA_PROB_APPROACH", "gumbel")
            if approach not in _GATER_MODES:
                raise ValueError(f"MGA_PROB_APPROACH must be one of {set(_GATER_MODES)}, got {approach}")
            self.gater = ProbMaskGater(mode=approach, tau=1.0, p_min=0.0, threshold=0.5, seed=None)

    # ------------------------------------------------------------------ reference surface
    @property
    def alpha(self) -> torch.Tensor:
        return F.softplus(self.beta)

    @property
    def hidden(self) -> int:
        return self.cam_mlp._modules["0"].out_features

    def block_config(self) -> BlockConfig:
        key = (self.hidden, self.k, bool(self.use_sigmoid_mask), float(self.tiny_thr), float(self.eps))
        cached = self.__dict__.get("_cfg_cache")
        if cached is None or cached[0] != key:                 # attributes are plain and may be edited after construction
            cached = (key, BlockConfig(*key))
            self.__dict__["_cfg_cache"] = cached
        return cached[1]

    def block_params(self):
        l1, l2 = self.cam_mlp._modules["0"], self.cam_mlp._modules["2"]     # (Sequential.__getitem__ is slow on the hot path)
        return (l1.weight, l1.bias, l2.weight, l2.bias, self.sam_conv.weight, self.beta)

    def forward(self, x: Union[torch.Tensor, Sequence[torch.Tensor]]) -> torch.Tensor:
        if isinstance(x, (list, tuple)):
            assert len(x) == 2, "MaskCBAM expects [feature, mask]"
            feat, mask = x
        else:
            feat, mask = x, None
        assert isinstance(feat, torch.Tensor) and feat.dim() == 4
        if os.getenv("MGA_PROB_MODE", False) and mask is not None:              # masked_cbam.py:163-164
            mask = self.gater(mask)
        if feat.is_cuda:
            params = tuple(p.float() if p.dtype != torch.float32 else p for p in self.block_params())
            return mask_cbam(feat, mask, *params, self.block_config())
        return _host_forward(feat, mask, self.block_params(), self.block_config())

    def extra_repr(self) -> str:
        return (f"channels={self.C}, r={self.r}, spatial_k={self.k}, use_sigmoid_mask={self.use_sigmoid_mask}, "
                f"tiny_mask_thr={self.tiny_thr}, eps={self.eps}, alpha={self.alpha.item():.4f}")


def _host_forward(x: torch.Tensor, mask: Optional[torch.Tensor], params, cfg: BlockConfig) -> torch.Tensor:
    """Same mathematics in differentiable PyTorch ops for HOST tensors only (build-time stride probe, device='cpu' runs)."""
    w1, b1, w2, b2, wsa, beta = params
    B, Cc, H, W = x.shape
    flat = x.flatten(2)                                                        # (B,C,N)
    gap = flat.mean(dim=2)
    if mask is None:
        avg, mx = gap, flat.max(dim=2)[0]                                      # index path: the first maximum gets the gradient
        plane = x.new_zeros(B, 1, H, W)
    else:
        m = mask.unsqueeze(1) if mask.dim() == 3 else mask
        if tuple(m.shape) != (B, 1, H, W):
            raise RuntimeError(f"mask shape {tuple(mask.shape)} does not match feature (B,1,H,W)=({B},1,{H},{W})")
        s = (m.sigmoid() if cfg.use_sigmoid_mask else m).to(x.dtype)
        sf = s.flatten(2)                                                      # (B,1,N)
        total = sf.sum(dim=2)                                                  # (B,1)
        use = (total / (H * W) >= cfg.tiny_thr).to(x.dtype)
        mavg = (flat * sf).sum(dim=2) / total.clamp_min(cfg.eps)
        avg = mavg * use + gap * (1.0 - use)
        low = torch.finfo(x.dtype).min
        mmax = flat.masked_fill(~(sf > 0.5), low).max(dim=2)[0]
        mx = torch.where(torch.isclose(mmax, torch.full_like(mmax, low)), gap, mmax)
        plane = s
    gate = lambda d: F.linear(F.relu(F.linear(d, w1.to(d.dtype), b1.to(d.dtype))), w2.to(d.dtype), b2.to(d.dtype))
    ca = torch.sigmoid(gate(avg) + gate(mx)).view(B, Cc, 1, 1)
    u = x * ca
    planes = torch.cat([u.max(dim=1, keepdim=True)[0], u.mean(dim=1, keepdim=True), plane], dim=1)
    sa = torch.sigmoid(F.conv2d(planes, wsa.to(x.dtype), padding=cfg.k // 2))
    return x + F.softplus(beta).to(x.dtype) * (u * sa - x)


# ---------------------------------------------------------------------------------------------------------
# MaskECA: mirror of mga_yolo/nn/modules/masked_eca.py:68-196
# ---------------------------------------------------------------------------------------------------------
def eca_kernel_size(channels: int, gamma: float = 2.0, b: float = 1.0, k_min: int = 3, k_max: int = 15) -> int:
    """Adaptive odd 1-D kernel size from the channel count (masked_eca.py:44-54)."""
    if channels <= 0:
        return k_min
    k = int(abs((channels.bit_length() - 1) / gamma + b))
    k = max(k_min, min(k, k_max))
    return k if k % 2 == 1 else k + 1


class _EcaCfg:
    """Attribute bag with the reference's ``cfg`` field names (masked_eca.py:57-65)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class MaskECA(nn.Module):
    """Mask-guided Efficient Channel Attention: masked average pooling (GAP fallback for tiny masks) -> conv1d over the
    channel axis -> sigmoid -> ``x * (1 + softplus(beta) * (w - 0.5))``.  Same constructor, parameter names
    (``conv1d.weight`` (1,1,k), ``beta``), ``alpha`` / ``scale_name`` / ``cfg`` attributes and ``[feat, mask]`` input as the
    reference class; device tensors run the HIP kernels (2 launches forward, 2 backward), host tensors plain PyTorch."""

    def __init__(self, channels: int, gamma: float = 2.0, b: float = 1.0, k_min: int = 3, k_max: int = 15,
                 use_sigmoid_mask: bool = True, tiny_mask_threshold: float = 1e-4, eps: float = 1e-6) -> None:
        super().__init__()
        self.cfg = _EcaCfg(channels=channels, gamma=gamma, b=b, k_min=k_min, k_max=k_max, use_sigmoid_mask=use_sigmoid_mask,
                           tiny_mask_threshold=tiny_mask_threshold, eps=eps)
        k = eca_kernel_size(channels, gamma=gamma, b=b, k_min=k_min, k_max=k_max)
        self.conv1d = nn.Conv1d(1, 1, kernel_size=k, padding=k // 2, bias=False)
        self.beta = nn.Parameter(torch.tensor(0.0, dtype=torch.float32))
        self.scale_name = {256: "P3", 512: "P4", 1024: "P5"}.get(channels, f"C{channels}")

    @property
    def alpha(self) -> torch.Tensor:
        return F.softplus(self.beta)

    def _maybe_rebuild_conv(self, channels: int) -> None:      # masked_eca.py:123-137
        if channels == self.cfg.channels:
            return
        k = eca_kernel_size(channels, gamma=self.cfg.gamma, b=self.cfg.b, k_min=self.cfg.k_min, k_max=self.cfg.k_max)
        wt = self.conv1d.weight
        self.conv1d = nn.Conv1d(1, 1, kernel_size=k, padding=k // 2, bias=False).to(device=wt.device, dtype=wt.dtype)
        self.cfg.channels = channels

    def eca_config(self) -> EcaConfig:
        return EcaConfig(k=int(self.conv1d.weight.shape[-1]), use_sigmoid_mask=bool(self.cfg.use_sigmoid_mask),
                         tiny_thr=float(self.cfg.tiny_mask_threshold), eps=float(self.cfg.eps))

    def forward(self, x: Union[torch.Tensor, Sequence[torch.Tensor]]) -> torch.Tensor:
        if isinstance(x, (list, tuple)):
            assert len(x) == 2, "MaskECA expects [feature, mask] as inputs"
            feat, mask = x
        else:
            feat, mask = x, None
        assert isinstance(feat, torch.Tensor) and feat.dim() == 4, "feature must be (B,C,H,W)"
        self._maybe_rebuild_conv(feat.shape[1])
        if feat.is_cuda:
            return mask_eca(feat, mask, self.conv1d.weight, self.beta, self.eca_config())
        return _eca_host_forward(feat, mask, self.conv1d.weight, self.beta, self.eca_config())

    def extra_repr(self) -> str:
        c = self.cfg
        return (f"C={c.channels}, gamma={c.gamma}, b={c.b}, k_min={c.k_min}, k_max={c.k_max}, sigmoid_mask={c.use_sigmoid_mask}, "
                f"tiny_thr={c.tiny_mask_threshold}, alpha={float(self.alpha.detach())}, scale='{self.scale_name}'")


def _eca_host_forward(x, mask, w, beta, cfg: EcaConfig) -> torch.Tensor:
    """Plain-PyTorch statement for HOST tensors (build-time stride probe, device='cpu')."""
    B, Cc, H, W = x.shape
    gap = x.flatten(2).mean(dim=2)
    if mask is None:
        v = gap
    else:
        m = mask.unsqueeze(1) if mask.dim() == 3 else mask
        if tuple(m.shape) != (B, 1, H, W):
            raise RuntimeError(f"mask shape {tuple(mask.shape)} does not match feature (B,1,H,W)=({B},1,{H},{W})")
        s = (m.sigmoid() if cfg.use_sigmoid_mask else m).to(x.dtype).flatten(2)     # (B,1,N)
        total = s.sum(dim=2)
        valid = (total / (H * W) >= cfg.tiny_thr).to(x.dtype)
        masked = (x.flatten(2) * s).sum(dim=2) / total.clamp_min(cfg.eps)
        v = masked * valid + gap * (1.0 - valid)
    gate = torch.sigmoid(F.conv1d(v.unsqueeze(1).to(w.dtype), w, padding=cfg.k // 2).squeeze(1)).view(B, Cc, 1, 1)
    return x * (1.0 + F.softplus(beta).to(gate.dtype) * (gate - 0.5)).to(x.dtype)


# ---------------------------------------------------------------------------------------------------------
# MGAMaskHead: mirror of mga_yolo/nn/modules/segmentation.py:34-127
# ---------------------------------------------------------------------------------------------------------
_HEAD_MAX_W = 500    # widest image row the mask-head 3x3 kernels hold in one pixel run (csrc/head.cuh: head_out_shape); wider features run torch ops


class _HeadCfg:
    """Attribute bag with the reference dataclass's field names (segmentation.py:35-53)."""

    def __init__(self, in_channels, hidden_channels, out_channels, norm, act, dropout):
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, hidden_channels, out_channels
        self.norm, self.act, self.dropout = norm, act, dropout


class ChannelLastLayerNorm(nn.Module):
    """LayerNorm over the channel axis of an NCHW tensor (segmentation.py:114-126); only built for norm="ln"."""

    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.ln = nn.LayerNorm(num_channels, eps=eps)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.ln(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)


class MGAMaskHead(nn.Module):
    """Coarse mask head of one pyramid level: Conv1x1(in -> hidden, no bias) -> BatchNorm2d -> SiLU -> Conv3x3(hidden -> 1) + bias,
    producing the LOGITS MaskCBAM takes as its mask and the segmentation loss is computed on.  Drop-in for the reference class:
    same constructor, same state_dict (``proj.0.weight``, ``proj.1.{weight,bias,running_mean,running_var,num_batches_tracked}``,
    ``head.{weight,bias}``), same initialisation order (kaiming_normal fan_out for the convs, BatchNorm ones / zeros), ``cfg`` and
    ``extra_repr``.  The containers hold the parameters; for device tensors in the configuration every reference YAML uses
    (norm="bn", act=SiLU, dropout=0, out_channels=1) the math runs in the HIP kernels (the 1x1 conv and its backward products on the
    matrix cores), BatchNorm's running statistics updated in place as torch does.  Host tensors and other constructor variants run
    the containers' own torch ops."""

    def __init__(self, in_channels: int, hidden_channels: int, out_channels: int = 1, norm: Optional[str] = "bn",
                 act: type = nn.SiLU, dropout: float = 0.0) -> None:
        super().__init__()
        self.cfg = _HeadCfg(in_channels, hidden_channels, out_channels, norm, act, dropout)
        layers = [nn.Conv2d(in_channels, hidden_channels, kernel_size=1, bias=False)]
        if norm == "bn":
            layers.append(nn.BatchNorm2d(hidden_channels))
        elif norm == "ln":
            layers.append(ChannelLastLayerNorm(hidden_channels))
        if act is not None:
            layers.append(act())
        if dropout and dropout > 0:
            layers.append(nn.Dropout2d(p=dropout))
        self.proj = nn.Sequential(*layers)
        self.head = nn.Conv2d(hidden_channels, out_channels, kernel_size=3, padding=1, bias=True)
        self._initialize()

    def _initialize(self) -> None:                              # segmentation.py:96-104 (same module traversal order)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def hip_path(self) -> bool:
        """True when the configuration is the one the HIP kernels implement."""
        c = self.cfg
        if not (c.norm == "bn" and c.act is nn.SiLU and not (c.dropout and c.dropout > 0) and c.out_channels == 1):
            return False
        bn = self.proj._modules["1"]
        return bool(bn.affine and bn.track_running_stats and bn.momentum is not None and bn.running_mean is not None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:         # (B, C_in, H, W) -> (B, C_out, H, W) logits
        if x.is_cuda and self.hip_path() and x.shape[-1] <= _HEAD_MAX_W:
            conv, bn = self.proj._modules["0"], self.proj._modules["1"]
            rm, rv = bn.running_mean, bn.running_var
            if rm.dtype != torch.float32 or rv.dtype != torch.float32:
                # a halved module: the validator calls model.half() on the EMA copy whenever AMP is on (U/engine/validator.py:147-149),
                # and half=True predict does the same.  The kernels keep statistics in fp32: in eval mode nothing is written back, so
                # fp32 copies serve; a TRAINING module with half-precision buffers (not something the reference trainer produces)
                # runs the containers' torch ops, which update the buffers in their own precision
                if self.training:
                    return self.head(self.proj(x))
                rm, rv = rm.float(), rv.float()
            return mask_head(x, conv.weight, bn.weight, bn.bias, rm, rv, bn.num_batches_tracked,
                             self.head.weight, self.head.bias, eps=bn.eps, momentum=bn.momentum, training=self.training)
        if x.is_cuda and self.hip_path():                         # a shape outside the kernels' range (rows wider than _HEAD_MAX_W)
            return self.head(self.proj(x))
        if x.is_cuda and not self.__dict__.get("_warned_variant"):
            import warnings
            warnings.warn("MGAMaskHead: constructor variant outside the HIP path (norm / act / dropout / out_channels); running torch ops")
            self.__dict__["_warned_variant"] = True
        return self.head(self.proj(x))

    def extra_repr(self) -> str:
        c = self.cfg
        return (f"in={c.in_channels}, hidden={c.hidden_channels}, out={c.out_channels}, norm={c.norm}, "
                f"act={c.act.__name__ if c.act else None}, dropout={c.dropout}")
