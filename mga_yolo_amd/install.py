"""Drop-in registration: make the reference engine build THIS MaskCBAM.

``parse_model`` resolves the YAML string "MaskCBAM" with ``globals()[m]`` inside its own module and picks the
``[feat, mask]`` branch by identity ``m is MaskCBAM`` against the same module global (U/nn/tasks.py:1676-1682, 1733-1739), so
rebinding that global -- before ``MGAModel(...)`` / ``YOLO(...)`` is constructed -- is enough.  The reference holds TWO
module objects made from that one file: ``ultralytics.nn.tasks`` (the vendored tree is put on sys.path,
mga_yolo/__init__.py:16-41) and ``mga_yolo.external.ultralytics.ultralytics.nn.tasks`` -- ``MGAModel`` derives from the
``DetectionModel`` of the SECOND (mga_yolo/model/model.py:10), so both are patched: every loaded module of the reference
that binds one of the class names to a class of that name is rebound.  The trainer's alpha logger finds the blocks by
``isinstance`` against ``mga_yolo.nn.modules.masked_cbam.MaskCBAM`` imported at call time (mga_yolo/model/trainer.py:286-295),
``MGAModel.init_criterion`` imports the loss classes at call time (model/model.py:103-117): covered by the same rule.
Nothing else in the reference changes.  See INTEGRATION.md; ``oracle/check_dropin.py`` runs this against the real factory.
"""
from __future__ import annotations

import importlib
import sys
from typing import Dict, List

from .module import MGAMaskHead, MaskCBAM, MaskECA
from .segloss import SegLossConfig, SegmentationLoss

# modules imported (if importable) before the scan, so that install() may run before OR after the reference is imported
_PRELOAD = ("ultralytics.nn.tasks", "mga_yolo.external.ultralytics.ultralytics.nn.tasks",
            "mga_yolo.nn.modules.masked_cbam", "mga_yolo.nn.modules.masked_eca", "mga_yolo.nn.modules.segmentation",
            "mga_yolo.nn.losses.segmentation", "mga_yolo.model.model")
_PREFIXES = ("ultralytics", "mga_yolo")
_TASKS_SUFFIX = "ultralytics.nn.tasks"
# name -> replacement.  parse_model treats MaskCBAM and MaskECA through the same branch (U/nn/tasks.py:1733)
_CLASSES: Dict[str, type] = {"MaskCBAM": MaskCBAM, "MaskECA": MaskECA, "MGAMaskHead": MGAMaskHead,
                             "SegmentationLoss": SegmentationLoss, "SegLossConfig": SegLossConfig}


def register(name: str, cls: type) -> None:
    """Add a (reference class name -> replacement) pair for later install() calls (used by optional rows, e.g. MGAMaskHead)."""
    _CLASSES[name] = cls


def _is_reference_module(name: str) -> bool:
    return any(name == p or name.startswith(p + ".") for p in _PREFIXES) and not name.startswith("mga_yolo_amd")


def install(strict: bool = False) -> List[str]:
    """Rebind the block classes in every loaded reference module that exports them.  Returns the module names patched.
    ``strict=True`` raises unless the module whose ``parse_model`` builds ``MGAModel`` (the one providing
    ``mga_yolo.model.model.DetectionModel``; ``ultralytics.nn.tasks`` when mga_yolo is absent) was patched."""
    for name in _PRELOAD:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                pass
    patched = []
    for name, mod in list(sys.modules.items()):
        if mod is None or not _is_reference_module(name):
            continue
        is_tasks = name.endswith(_TASKS_SUFFIX)
        hit = False
        for cls_name, cls in _CLASSES.items():
            cur = mod.__dict__.get(cls_name, _MISSING) if hasattr(mod, "__dict__") else _MISSING
            if cur is cls:
                hit = True
                continue
            # a class of that name (the reference's own), or the `= None` left by tasks.py's guarded import (U/nn/tasks.py:72-90)
            if (isinstance(cur, type) and cur.__name__ == cls_name) or (is_tasks and cur is None and cls_name in ("MaskCBAM", "MaskECA", "MGAMaskHead")):
                _UNDO.append((name, cls_name, cur))
                setattr(mod, cls_name, cls)
                hit = True
        if hit:
            patched.append(name)
    if strict:
        need = _factory_module_name()
        if need not in patched:
            raise RuntimeError(f"{need} (the module whose parse_model builds the model) could not be patched: "
                               f"patched = {patched}")
    return sorted(patched)


_MISSING = object()
_UNDO: list = []          # (module name, attribute, previous value) of every rebinding install() made


def uninstall() -> int:
    """Put back what install() replaced (A/B runs against the reference's own classes).  Returns the number of names restored."""
    n = 0
    while _UNDO:
        name, attr, old = _UNDO.pop()
        mod = sys.modules.get(name)
        if mod is not None:
            setattr(mod, attr, old)
            n += 1
    return n


def _factory_module_name() -> str:
    mm = sys.modules.get("mga_yolo.model.model")
    if mm is None:
        try:
            mm = importlib.import_module("mga_yolo.model.model")
        except Exception:
            mm = None
    dm = getattr(mm, "DetectionModel", None)
    return dm.__module__ if isinstance(dm, type) else "ultralytics.nn.tasks"
