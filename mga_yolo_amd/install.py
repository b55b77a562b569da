"""Drop-in registration: make the reference engine build THIS MaskCBAM.

``parse_model`` resolves the YAML string "MaskCBAM" with ``globals()[m]`` inside ``ultralytics.nn.tasks`` and picks the
``[feat, mask]`` branch by identity ``m is MaskCBAM`` against the same module global (U/nn/tasks.py:1676-1682, 1733-1739),
so rebinding that one global -- before ``YOLO(...)`` / ``MGAModel(...)`` is constructed -- is enough; the trainer's alpha
logger finds the module by ``isinstance`` against ``mga_yolo.nn.modules.masked_cbam.MaskCBAM`` (mga_yolo/model/trainer.py:286-295),
so that name is rebound as well.  Nothing else in the reference changes.  See INTEGRATION.md.
"""
from __future__ import annotations

import importlib
import sys
from typing import List

from .module import MaskCBAM, MaskECA
from .segloss import SegLossConfig, SegmentationLoss

_TARGETS = ("ultralytics.nn.tasks", "ultralytics.nn", "ultralytics.nn.modules",
            "mga_yolo.nn.modules.masked_cbam", "mga_yolo.nn.modules.masked_eca", "mga_yolo.nn.modules", "mga_yolo.nn")
_CLASSES = {"MaskCBAM": MaskCBAM, "MaskECA": MaskECA}      # parse_model treats both through the same branch (U/nn/tasks.py:1733)
# MGAModel.init_criterion imports these two names from this module at call time (mga_yolo/model/model.py:103-117)
_LOSS_TARGET = "mga_yolo.nn.losses.segmentation"
_LOSS_CLASSES = {"SegmentationLoss": SegmentationLoss, "SegLossConfig": SegLossConfig}


def install(strict: bool = False) -> List[str]:
    """Rebind ``MaskCBAM`` in every reference module that exports it.  Returns the module names patched.
    ``strict=True`` raises if the parse_model namespace (ultralytics.nn.tasks) could not be patched."""
    patched = []
    for name in _TARGETS:
        mod = sys.modules.get(name)
        if mod is None:
            try:
                mod = importlib.import_module(name)
            except Exception:
                continue
        hit = False
        for cls_name, cls in _CLASSES.items():
            if hasattr(mod, cls_name) or name == "ultralytics.nn.tasks":
                setattr(mod, cls_name, cls)
                hit = True
        if hit:
            patched.append(name)
    mod = sys.modules.get(_LOSS_TARGET)
    if mod is None:
        try:
            mod = importlib.import_module(_LOSS_TARGET)
        except Exception:
            mod = None
    if mod is not None:
        for cls_name, cls in _LOSS_CLASSES.items():
            setattr(mod, cls_name, cls)
        patched.append(_LOSS_TARGET)
    if strict and "ultralytics.nn.tasks" not in patched:
        raise RuntimeError("ultralytics.nn.tasks is not importable: nothing to install into")
    return patched
