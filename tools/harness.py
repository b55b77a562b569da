"""Train-step harness around the hot path -- SURVEY 8d images/s definition (2): a self-contained YOLOv8-SHAPED backbone stand-in, the three
MGA levels (MGAMaskHead -> [feat, mask] -> MaskCBAM, mga_yolo/model/model.py:57-74), a stand-in detection loss, the multi-scale
segmentation loss and the Kendall combine (model.py:196-206), wrapped in the reference's own
``DistributedDataParallel(find_unused_parameters=True)`` (U/engine/trainer.py:366-367) when there is more than one rank.

The stand-in is NOT a port of the reference's backbone (C2f / SPPF / Detect are out of scope): a plain ``torch.nn``
Conv-BatchNorm-SiLU stride stack with YOLOv8's channel widths at the scale asked for, producing P3 / P4 / P5 at strides 8 / 16 / 32.
Its only job is to give the step a realistic amount of out-of-scope work and of out-of-scope gradients (DDP all-reduces ALL of them)
around the in-scope layers, which are this package's modules: on a GPU they run the HIP kernels, on the host their torch statement.

Used by bench.py (extra key ``train_harness``) and tests/test_dp_gloo.py (2 gloo ranks on CPU)."""
from __future__ import annotations

import os
import sys
import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# P3 / P4 / P5 channel widths of yolov8_cbam.yaml per scale (measured by building MGAModel)
WIDTHS = {"n": (64, 128, 256), "s": (128, 256, 512), "m": (256, 512, 512), "l": (256, 512, 512), "x": (384, 768, 768)}


def _cbs(cin: int, cout: int, stride: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout), nn.SiLU(inplace=True))


class HarnessModel(nn.Module):
    """image (B,3,S,S) -> refined P3/P4/P5 features (what Detect would take) + the three mask-logit maps."""

    def __init__(self, scale: str = "n", depth: int = 3):      # depth 3: 2.75 M parameters at n (YOLOv8n+MGA: 2.96 M), 11.0 M at s (10.9 M)
        super().__init__()
        from mga_yolo_amd import MGAMaskHead, MaskCBAM
        c3, c4, c5 = WIDTHS[scale]
        stem = max(16, c3 // 4)
        self.stem = nn.Sequential(_cbs(3, stem, 2), _cbs(stem, 2 * stem, 2))                   # /4
        self.to_p3 = nn.Sequential(_cbs(2 * stem, c3, 2), *[_cbs(c3, c3, 1) for _ in range(depth)])      # /8
        self.to_p4 = nn.Sequential(_cbs(c3, c4, 2), *[_cbs(c4, c4, 1) for _ in range(depth)])             # /16
        self.to_p5 = nn.Sequential(_cbs(c4, c5, 2), *[_cbs(c5, c5, 1) for _ in range(depth)])             # /32
        self.heads = nn.ModuleList([MGAMaskHead(c, max(8, c // 4)) for c in (c3, c4, c5)])     # yolov8_cbam.yaml: hidden = C / 4
        self.blocks = nn.ModuleList([MaskCBAM(c) for c in (c3, c4, c5)])
        self.mtl_log_vars = nn.Parameter(torch.zeros(2))                                        # model.py:119-121
        from mga_yolo_amd import SegLossConfig, SegmentationLoss
        self.criterion = SegmentationLoss(SegLossConfig())

    def features(self, img: torch.Tensor) -> Tuple[List[torch.Tensor], Dict[str, torch.Tensor]]:
        p3 = self.to_p3(self.stem(img))
        p4 = self.to_p4(p3)
        p5 = self.to_p5(p4)
        refined, seg = [], {}
        for key, f, head, blk in zip(("p3", "p4", "p5"), (p3, p4, p5), self.heads, self.blocks):
            m = head(f)                                            # model.py:57-64: the layer loop's list input
            seg[key] = m
            refined.append(blk([f, m]))
        return refined, seg

    def forward(self, img: torch.Tensor, masks: Optional[List[torch.Tensor]] = None):
        """With masks: the training loss, formed INSIDE forward as the reference does (the trainer calls model(batch), which returns
        model.loss(batch): under DDP every parameter -- the Kendall log-variances included -- must be used inside the wrapped forward)."""
        refined, seg = self.features(img)
        if masks is None:
            return refined, seg
        from mga_yolo_amd import kendall_combine
        det = torch.stack([r.float().pow(2).mean() for r in refined])         # stand-in detection 3-vector (box, cls, dfl)
        seg_total, _ = self.criterion(seg, masks)                              # model.py:196-202
        return kendall_combine(det, seg_total, self.mtl_log_vars).sum()        # model.py:204-206; trainer: loss.sum().backward()


def synthetic_batch(batch: int, size: int, device, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, size, size, generator=g).to(device)
    masks = [(torch.rand(batch, 1, size // s, size // s, generator=g) > 0.9).float().to(device) for s in (8, 16, 32)]
    return img, masks


def build(scale: str, device, world: int = 1, depth: int = 3):
    torch.manual_seed(0)
    model = HarnessModel(scale, depth).to(device)
    if device.type == "cuda":
        model = model.to(memory_format=torch.channels_last)
    wrapped = model
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        wrapped = DDP(model, device_ids=[device.index] if device.type == "cuda" else None, find_unused_parameters=True)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
    return model, wrapped, opt


def train_step(wrapped, opt, img, masks, amp: Optional[torch.dtype] = None, scaler=None) -> torch.Tensor:
    opt.zero_grad(set_to_none=True)
    if amp is not None:
        with torch.autocast(img.device.type, dtype=amp):
            loss = wrapped(img, masks)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
    else:
        loss = wrapped(img, masks)
        loss.backward()
        opt.step()
    return loss.detach()


def run(scale: str = "n", batch: int = 32, size: int = 640, steps: int = 10, warmup: int = 3, device=None, world: int = 1, rank: int = 0,
        amp: Optional[str] = None) -> dict:
    """Time `steps` train steps; returns images/s over all ranks (max elapsed over ranks when world > 1)."""
    import torch.distributed as dist
    device = device or torch.device("cuda", 0)
    amp_dt = {None: None, "fp16": torch.float16, "bf16": torch.bfloat16}[amp]
    model, wrapped, opt = build(scale, device, world)
    scaler = torch.amp.GradScaler(device.type, enabled=amp_dt is torch.float16) if amp_dt is not None else None
    img, masks = synthetic_batch(batch, size, device, seed=100 + rank)
    if device.type == "cuda":
        img = img.contiguous(memory_format=torch.channels_last)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    for _ in range(warmup):
        train_step(wrapped, opt, img, masks, amp_dt, scaler)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = train_step(wrapped, opt, img, masks, amp_dt, scaler)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    nparam = sum(p.numel() for p in model.parameters())
    return dict(images_per_s=round(world * batch * steps / el, 1), ms_per_step=round(el * 1e3 / steps, 3), steps=steps, warmup=warmup,
                batch_per_gpu=batch, image=size, scale=scale, amp=amp, loss=float(loss), parameters=nparam, grad_bytes_per_step=4 * nparam,
                ddp=bool(world > 1), note="plain Conv-BN-SiLU stride stack with YOLOv8 widths (stand-in for the out-of-scope backbone) -> "
                "MGAMaskHead / MaskCBAM / SegmentationLoss / Kendall combine of this package; fwd + loss + bwd + SGD step, eager autograd")


if __name__ == "__main__":
    import json
    print(json.dumps(run(sys.argv[1] if len(sys.argv) > 1 else "n", steps=int(sys.argv[2]) if len(sys.argv) > 2 else 10)))
