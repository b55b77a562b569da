"""TEST INFRASTRUCTURE -- CPU restatement of the reference's MGAMaskHead, default configuration
(norm="bn", act=SiLU, dropout=0): mga_yolo/nn/modules/segmentation.py:56-110

    proj = Conv2d(C, hid, 1, bias=False) -> BatchNorm2d(hid) -> SiLU          (:79-90)
    head = Conv2d(hid, out, 3, padding=1, bias=True)                          (:92)
    forward(x) = head(proj(x))  -> mask LOGITS (B, out, H, W)                  (:106-109)

Explicit forward that returns every intermediate and a hand-derived backward (what autograd does for the three modules), in
torch fp32 / fp64.  Only tests/, __graft_entry__.smoke() and the benchmarks' CPU-baseline legs may import this.

PINNED by outputs of the reference itself: ``oracle/gen_golden_head.py`` imported the reference class in the build container and
stored inputs, parameters, logits, updated BatchNorm running statistics and every gradient in ``tests/golden/head_*.npz``
(train and eval mode, eps / momentum as Ultralytics' initialize_weights sets them, U/utils/torch_utils.py:564-574, and torch's
defaults); ``tests/test_maskhead.py`` checks this restatement against them.

BatchNorm2d in training mode (torch semantics): per channel j over (B,H,W), n = B*H*W
    mu = mean z, var = biased variance, zhat = (z - mu) / sqrt(var + eps), a = gamma * zhat + beta
    running_mean <- (1-m) running_mean + m mu ; running_var <- (1-m) running_var + m var * n/(n-1) ; num_batches_tracked += 1
in eval mode mu / var are the running statistics and nothing is updated.
Backward, with g = dL/dlogits (B,out,H,W):
    g_s   = conv_transpose3x3(g, W_h)                   gW_h[o,j,u,v] = sum g[b,o,y,x] s[b,j,y+u-1,x+v-1]      gb_h[o] = sum g
    g_a   = g_s * silu'(a),  silu'(a) = sig(a) (1 + a (1 - sig(a)))
    ggamma= sum g_a zhat      gbeta = sum g_a
    g_z   = gamma * rstd * (g_a - gbeta/n - zhat * ggamma/n)        (training)        g_z = gamma * rstd * g_a   (eval)
    g_x[b,c,hw] = sum_j W1[j,c] g_z[b,j,hw]             gW1[j,c] = sum_{b,hw} g_z[b,j,hw] x[b,c,hw]
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class HeadParams:
    w1: torch.Tensor            # proj.0.weight (hid, C, 1, 1) stored as (hid, C)
    gamma: torch.Tensor         # proj.1.weight (hid)
    beta: torch.Tensor          # proj.1.bias (hid)
    running_mean: torch.Tensor  # proj.1.running_mean (hid)
    running_var: torch.Tensor   # proj.1.running_var (hid)
    wh: torch.Tensor            # head.weight (out, hid, 3, 3)
    bh: torch.Tensor            # head.bias (out)
    eps: float = 1e-5
    momentum: float = 0.1

    @staticmethod
    def from_state_dict(sd: Dict[str, torch.Tensor], eps: float = 1e-5, momentum: float = 0.1) -> "HeadParams":
        w1 = sd["proj.0.weight"]
        return HeadParams(w1.reshape(w1.shape[0], w1.shape[1]).clone(), sd["proj.1.weight"].clone(), sd["proj.1.bias"].clone(),
                          sd["proj.1.running_mean"].clone(), sd["proj.1.running_var"].clone(), sd["head.weight"].clone(),
                          sd["head.bias"].clone(), eps, momentum)

    @staticmethod
    def default_init(C: int, hid: int, out: int = 1, seed: int = 0, eps: float = 1e-5, momentum: float = 0.1) -> "HeadParams":
        """segmentation.py:96-104: kaiming_normal_(fan_out, relu) for both convs in creation order, zero bias, BN ones/zeros."""
        torch.manual_seed(seed)
        w1 = torch.empty(hid, C, 1, 1)
        wh = torch.empty(out, hid, 3, 3)
        torch.nn.init.kaiming_normal_(w1, mode="fan_out", nonlinearity="relu")
        torch.nn.init.kaiming_normal_(wh, mode="fan_out", nonlinearity="relu")
        return HeadParams(w1.reshape(hid, C), torch.ones(hid), torch.zeros(hid), torch.zeros(hid), torch.ones(hid), wh, torch.zeros(out),
                          eps, momentum)


@dataclass
class HeadCtx:
    z: torch.Tensor
    mean: torch.Tensor
    var: torch.Tensor
    rstd: torch.Tensor
    zhat: torch.Tensor
    a: torch.Tensor
    s: torch.Tensor
    new_running_mean: torch.Tensor
    new_running_var: torch.Tensor


def forward(x: torch.Tensor, p: HeadParams, training: bool = True) -> Tuple[torch.Tensor, HeadCtx]:
    B, C, H, W = x.shape
    hid = p.w1.shape[0]
    xf = x.float()
    z = torch.einsum("jc,bchw->bjhw", p.w1, xf)                                   # segmentation.py:81 (1x1 conv, no bias)
    n = B * H * W
    if training:                                                                  # segmentation.py:83 (BatchNorm2d, batch statistics)
        mean = z.mean(dim=(0, 2, 3))
        var = z.var(dim=(0, 2, 3), unbiased=False)
        unb = var * (n / (n - 1)) if n > 1 else var
        new_rm = (1 - p.momentum) * p.running_mean + p.momentum * mean
        new_rv = (1 - p.momentum) * p.running_var + p.momentum * unb
    else:
        mean, var = p.running_mean, p.running_var
        new_rm, new_rv = p.running_mean, p.running_var
    rstd = torch.rsqrt(var + p.eps)
    zhat = (z - mean.view(1, hid, 1, 1)) * rstd.view(1, hid, 1, 1)
    a = zhat * p.gamma.view(1, hid, 1, 1) + p.beta.view(1, hid, 1, 1)
    s = a * torch.sigmoid(a)                                                      # segmentation.py:87 (SiLU)
    logits = F.conv2d(s, p.wh, p.bh, padding=1)                                   # segmentation.py:92
    return logits, HeadCtx(z, mean, var, rstd, zhat, a, s, new_rm, new_rv)


def backward(g: torch.Tensor, x: torch.Tensor, p: HeadParams, c: HeadCtx, training: bool = True) -> Dict[str, Optional[torch.Tensor]]:
    B, C, H, W = x.shape
    hid = p.w1.shape[0]
    n = B * H * W
    g = g.float()
    g_s = F.conv_transpose2d(g, p.wh, padding=1)
    sp = F.pad(c.s, (1, 1, 1, 1))
    gwh = torch.zeros_like(p.wh)
    for u in range(3):
        for v in range(3):
            gwh[:, :, u, v] = torch.einsum("bohw,bjhw->oj", g, sp[:, :, u:u + H, v:v + W])
    gbh = g.sum(dim=(0, 2, 3))
    sig = torch.sigmoid(c.a)
    g_a = g_s * (sig * (1 + c.a * (1 - sig)))
    ggamma = (g_a * c.zhat).sum(dim=(0, 2, 3))
    gbeta = g_a.sum(dim=(0, 2, 3))
    k = (p.gamma * c.rstd).view(1, hid, 1, 1)
    if training:
        g_z = k * (g_a - gbeta.view(1, hid, 1, 1) / n - c.zhat * ggamma.view(1, hid, 1, 1) / n)
    else:
        g_z = k * g_a
    gx = torch.einsum("jc,bjhw->bchw", p.w1, g_z)
    gw1 = torch.einsum("bjhw,bchw->jc", g_z, x.float())
    return dict(gx=gx, gw1=gw1, ggamma=ggamma, gbeta=gbeta, gwh=gwh, gbh=gbh)


def reference_form_step(x, p: HeadParams, g, training: bool = True):
    """The same computation issued as the reference issues it (three nn modules' ATen ops through autograd): the CPU baseline."""
    hid, C = p.w1.shape
    xl = x.clone().requires_grad_(True)
    ws = [t.clone().requires_grad_(True) for t in (p.w1.reshape(hid, C, 1, 1), p.gamma, p.beta, p.wh, p.bh)]
    rm, rv = p.running_mean.clone(), p.running_var.clone()
    z = F.conv2d(xl, ws[0])
    a = F.batch_norm(z, rm, rv, ws[1], ws[2], training, p.momentum, p.eps)
    y = F.conv2d(F.silu(a), ws[3], ws[4], padding=1)
    y.backward(g)
    return y.detach(), xl.grad
