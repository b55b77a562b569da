"""Generate the golden vectors that pin ``oracle/segloss_oracle.py`` and the HIP segmentation-loss path.

TEST INFRASTRUCTURE.  Run ONCE in the build container, where the upstream reference is mounted read-only:

    PYTHONDONTWRITEBYTECODE=1 YOLO_CONFIG_DIR=/tmp/yolo_cfg python oracle/gen_golden_segloss.py

The reference's ``mga_yolo/nn/losses/segmentation.py`` imports Ultralytics' LOGGER at module scope (:7), whose package imports
``cv2`` (absent from this image): the in-process stand-in of SURVEY appendix A2 (``sys.modules["cv2"] = MagicMock()``; nothing of it
is executed on this path) lets the reference's own ``SegmentationLoss`` be imported and run.  Written:

* ``tests/golden/segloss_*.npz`` -- logits, targets, config, and the reference's total / log entries / d total / d logits for both
  modes (BCE + soft Dice; Unified Focal), equal-size / nearest-resized / bilinear-resized (``MGA_PROB_MODE``) targets, 3-D targets,
  missing levels, scale weights and ``loss_lambda``;
* ``tests/golden/kendall_*.npz``  -- the multi-task combine of ``MGAModel.loss`` (mga_yolo/model/model.py:196-206) evaluated by the
  reference model itself on a synthetic batch: detection-loss vector, segmentation total, the two log-variances, the combined
  loss and its gradient w.r.t. the log-variances.

Only data is written -- no reference source text.
"""
import importlib.metadata as md
import json
import os
import shutil
import sys
import tempfile
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np

cv2 = MagicMock(name="cv2"); cv2.__version__ = "4.10.0"; cv2.__spec__ = None
sys.modules["cv2"] = cv2
_real_version = md.version
md.version = lambda n: "0.25.0" if n == "torchvision" else _real_version(n)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

import mga_yolo  # noqa: E402,F401
from mga_yolo.nn.losses.segmentation import SegLossConfig, SegmentationLoss  # noqa: E402  (the reference itself)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_num_threads(1)

# name, B, pred sizes (p3,p4,p5), target size (None = same), cfg kwargs, extras
CASES = [
    ("base",        3, ((16, 16), (8, 8), (4, 4)), None,      dict(), dict()),
    ("weights",     2, ((16, 16), (8, 8), (4, 4)), None,      dict(scale_weights=(0.5, 2.0, 3.0), loss_lambda=0.25, bce_weight=0.7, dice_weight=1.3, smooth=0.5), dict()),
    ("nearest64",   2, ((16, 16), (8, 8), (4, 4)), (64, 64),  dict(), dict()),
    ("nearest_odd", 2, ((20, 12), (10, 6), (5, 3)), (17, 23), dict(scale_weights=(1.0, 0.5, 0.25)), dict()),
    ("bilinear64",  2, ((16, 16), (8, 8), (4, 4)), (64, 64),  dict(), dict(prob_mode=True, soft_targets=True)),
    ("bilinear_odd", 2, ((20, 12), (10, 6), (5, 3)), (17, 23), dict(loss_lambda=0.8), dict(prob_mode=True, soft_targets=True)),
    ("bilinear_up", 1, ((16, 16), (8, 8), (4, 4)), (5, 7),    dict(), dict(prob_mode=True, soft_targets=True)),
    ("target3d",    2, ((8, 8), (4, 4), (2, 2)),   None,      dict(), dict(target3d=True)),
    ("missing_p4",  2, ((16, 16), (8, 8), (4, 4)), None,      dict(), dict(drop=("p4",))),
    ("two_targets", 2, ((16, 16), (8, 8), (4, 4)), None,      dict(), dict(n_targets=2)),
    ("short_weights", 2, ((8, 8), (4, 4), (2, 2)), None,      dict(scale_weights=(0.3,)), dict()),
    ("ufl",         3, ((16, 16), (8, 8), (4, 4)), None,      dict(use_unified_focal=True), dict()),
    ("ufl_params",  2, ((16, 16), (8, 8), (4, 4)), (32, 32),  dict(use_unified_focal=True, ufl_lambda=0.3, ufl_delta=0.7, ufl_gamma=0.75, smooth=0.5,
                                                                   scale_weights=(1.0, 0.5, 0.25), loss_lambda=0.8), dict()),
    ("ufl_confident", 2, ((8, 8), (4, 4), (2, 2)), None,      dict(use_unified_focal=True), dict(logit_scale=12.0)),
    ("cfg2_masks",  4, ((80, 80), (40, 40), (20, 20)), None,  dict(), dict()),
    ("fp16",        2, ((16, 16), (8, 8), (4, 4)), (64, 64),  dict(), dict(dtype="float16")),
]


def run_case(name, B, sizes, tsize, ckw, ex):
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    dtype = getattr(torch, ex.get("dtype", "float32"))
    preds, tg = {}, []
    for k, (H, W) in zip(("p3", "p4", "p5"), sizes):
        preds[k] = (torch.randn(B, 1, H, W, generator=g) * ex.get("logit_scale", 2.0)).to(dtype)
        th, tw = (H, W) if tsize is None else tsize
        r = torch.rand(B, 1, th, tw, generator=g)
        t = r if ex.get("soft_targets") else (r > 0.7).float()
        tg.append(t.squeeze(1) if ex.get("target3d") else t)
    for k in ex.get("drop", ()):
        preds.pop(k)
    tg = tg[:ex.get("n_targets", 3)]
    if ex.get("prob_mode"):
        os.environ["MGA_PROB_MODE"] = "1"
    else:
        os.environ.pop("MGA_PROB_MODE", None)
    cfg = SegLossConfig(**ckw)
    crit = SegmentationLoss(cfg)
    leaf = {k: v.clone().float().requires_grad_(True) for k, v in preds.items()}     # fp16 case: the reference sees fp32 copies of the
    total, logs = crit({k: v for k, v in leaf.items()}, tg)                          # rounded logits (CPU half kernels differ from AMP)
    total.backward()
    os.environ.pop("MGA_PROB_MODE", None)
    arrays = {}
    for k, v in preds.items():
        arrays[f"logits.{k}"] = v.numpy()
        arrays[f"grad.{k}"] = (torch.zeros_like(leaf[k]) if leaf[k].grad is None else leaf[k].grad).numpy()   # level skipped -> no gradient
    for i, t in enumerate(tg):
        arrays[f"target.{i}"] = t.numpy()
    meta = dict(cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in ckw.items()}, prob_mode=bool(ex.get("prob_mode")),
                logs=logs, total=float(total), dtype=ex.get("dtype", "float32"))
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"segloss_{name}.npz"), **arrays)
    print(f"segloss {name:14s} total={float(total):.6f} keys={sorted(logs)}")


def kendall_cases():
    """MGAModel.loss on a synthetic batch: the reference model evaluates det + seg and combines them (model.py:196-206)."""
    from mga_yolo.model.model import MGAModel
    tmp = tempfile.mkdtemp(prefix="kendall_")
    try:
        yaml = os.path.join(tmp, "yolov8n_cbam.yaml")
        shutil.copy("/root/reference/configs/models/yolov8_cbam.yaml", yaml)
        torch.manual_seed(0)
        m = MGAModel(yaml, nc=1, verbose=False)
        m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
        m._ensure_criteria()
        for tag, lv in (("init", (0.0, 0.0)), ("trained", (0.4, -0.7))):
            with torch.no_grad():
                m.mtl_log_vars.copy_(torch.tensor(lv))
            g = torch.Generator().manual_seed(5)
            B = 2
            img = torch.rand(B, 3, 128, 128, generator=g)
            masks = [(torch.rand(B, 1, 128 // s, 128 // s, generator=g) > 0.8).float() for s in (8, 16, 32)]
            n = 4
            batch = {"img": img, "masks_multi": masks, "cls": torch.zeros(n, 1), "bboxes": torch.rand(n, 4, generator=g) * 0.4 + 0.3,
                     "batch_idx": torch.arange(n) % B}
            m.train()
            m.zero_grad()
            preds = m._predict_once(img)
            det_loss, _ = m.det_criterion(preds["det"], batch)
            seg_total, _ = m.seg_criterion(preds["seg"], masks)
            m.zero_grad()
            total, items = m.loss(batch, preds=preds)
            total.sum().backward()
            arrays = dict(det_loss=det_loss.detach().numpy(), seg_total=seg_total.detach().numpy(), log_vars=np.array(lv, dtype=np.float32),
                          total=total.detach().numpy(), g_log_vars=m.mtl_log_vars.grad.numpy(), loss_items=items.detach().numpy())
            np.savez_compressed(os.path.join(OUT, f"kendall_{tag}.npz"), **arrays)
            print(f"kendall {tag}: det={det_loss.detach().tolist()} seg={float(seg_total):.6f} total={total.detach().tolist()} "
                  f"g_log_vars={m.mtl_log_vars.grad.tolist()}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    os.makedirs(OUT, exist_ok=True)
    for c in CASES:
        run_case(*c)
    kendall_cases()


if __name__ == "__main__":
    sys.exit(main())
