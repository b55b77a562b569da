"""Drop-in proof against the reference's REAL model factory.

TEST INFRASTRUCTURE, build container only (the reference is mounted read-only at /root/reference and never travels):

    PYTHONDONTWRITEBYTECODE=1 YOLO_CONFIG_DIR=/tmp/yolo_cfg python oracle/check_dropin.py

What it does, in the order INTEGRATION.md prescribes for a user of the reference:
  1. imports the reference (``mga_yolo``, ``MGAModel``) -- cv2 / torchvision are absent from this image, so the two in-process
     stand-ins of SURVEY appendix A2 satisfy the module-scope imports (nothing of them is executed on this path);
  2. builds the reference's ``MGAModel`` from ``configs/models/yolov8_cbam.yaml`` (scales n and s) un-patched, at seed 0;
  3. calls ``mga_yolo_amd.install(strict=True)`` AFTER those imports and builds the same model again through the reference's own
     ``parse_model`` (U/nn/tasks.py:1676-1682, 1733-1739, 1763-1766);
  4. records: the classes of layers 22-27, the attributes parse_model attaches (``i f type np``), state_dict key/shape/value
     equality and cross-loading with ``strict=True``, eval- and train-mode CPU forward equality (``det`` and ``seg`` outputs), the
     multi-task loss and its backward (``MGAModel.loss``, model/model.py:123-214) on a synthetic batch, and the trainer's alpha
     collection rule (``isinstance`` + ``.alpha``, model/trainer.py:286-295).
The outcome is written to ``tests/golden/dropin_report.json`` (data only); ``tests/test_dropin_fixture.py`` asserts on it and
re-checks, without the reference, everything that can be re-checked from the recorded keys / shapes.
"""
import importlib.metadata as md
import json
import os
import shutil
import sys
import tempfile
from types import SimpleNamespace
from unittest.mock import MagicMock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "dropin_report.json")


def stub_missing_imports():
    cv2 = MagicMock(name="cv2")
    cv2.__version__ = "4.10.0"
    cv2.__spec__ = None
    sys.modules["cv2"] = cv2
    real = md.version
    md.version = lambda n: "0.25.0" if n == "torchvision" else real(n)


def synthetic_batch(torch, B, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, 256, 256, generator=g)
    masks = [(torch.rand(B, 1, 256 // s, 256 // s, generator=g) > 0.8).float() for s in (8, 16, 32)]
    n = 3 * B
    return {"img": img, "masks_multi": masks, "cls": torch.zeros(n, 1),
            "bboxes": torch.rand(n, 4, generator=g) * 0.4 + 0.3, "batch_idx": torch.arange(n) % B}


def main():
    stub_missing_imports()
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    import torch
    torch.set_num_threads(4)
    import mga_yolo  # noqa: F401  (puts the vendored ultralytics on sys.path, mga_yolo/__init__.py:16-32)
    from mga_yolo.model.model import MGAModel
    from mga_yolo.nn.modules.masked_cbam import MaskCBAM as RefCBAM
    from mga_yolo.nn.modules.segmentation import MGAMaskHead as RefHead

    tmp = tempfile.mkdtemp(prefix="dropin_")
    report = {"torch": torch.__version__, "scales": {}}
    try:
        import mga_yolo_amd
        for scale in ("n", "s"):
            yaml = os.path.join(tmp, f"yolov8{scale}_cbam.yaml")      # yaml_model_load takes the scale from the file name
            shutil.copy(os.path.join(REF, "configs", "models", "yolov8_cbam.yaml"), yaml)

            def build():
                torch.manual_seed(0)
                m = MGAModel(yaml, nc=1, verbose=False)
                m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
                return m

            def flat(o):
                if isinstance(o, torch.Tensor):
                    return [o.detach()]
                if isinstance(o, dict):
                    return [t for k in sorted(o) for t in flat(o[k])]
                if isinstance(o, (list, tuple)):
                    return [t for v in o for t in flat(v)]
                return []

            def rel(x, y):
                return float((x.double() - y.double()).abs().max() / y.double().abs().max().clamp_min(1e-30))

            def max_diff(fa, fb):
                assert len(fa) == len(fb)
                return max(rel(x, y) for x, y in zip(fa, fb))

            def run_all(m):
                """eval forward, train forward, multi-task loss + backward on the synthetic batch"""
                out = {}
                m.eval()
                with torch.no_grad():
                    out["eval"] = flat(m(batch["img"]))
                m.train()
                out["train"] = flat(m(batch["img"]))
                m.zero_grad()
                loss, _items = m.loss(batch)
                loss.sum().backward()
                out["loss"] = float(loss.sum())
                out["grads"] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
                out["crit"] = f"{type(m.seg_criterion).__module__}.{type(m.seg_criterion).__name__}"
                return out

            batch = synthetic_batch(torch, 2, seed=11)
            # ---- the reference, untouched: build, give it a perturbed ("trained") state, run everything -----------------
            ref = build()
            assert all(isinstance(ref.model[i], RefCBAM) for i in (23, 25, 27))
            sd_init = {k: v.clone() for k, v in ref.state_dict().items()}
            g = torch.Generator().manual_seed(7)
            trained = {k: (v + 0.05 * torch.randn(v.shape, generator=g)) if v.is_floating_point() else v for k, v in sd_init.items()}
            ref.load_state_dict(trained, strict=True)
            o_ref = run_all(ref)
            trained_after = {k: v.clone() for k, v in ref.state_dict().items()}     # (BN running statistics moved in train mode)
            # ---- install AFTER the reference imports, as a user would, and build through the reference's parse_model -------
            patched = mga_yolo_amd.install(strict=True)
            new = build()
            R = {"patched_modules": patched, "layers": {}}
            for i in range(22, 28):
                L = new.model[i]
                R["layers"][str(i)] = dict(cls=f"{type(L).__module__}.{type(L).__name__}", i=L.i, f=L.f, type=L.type, np=int(L.np),
                                           ref_np=int(ref.model[i].np),
                                           state={k: list(v.shape) for k, v in L.state_dict().items()})
            R["blocks_are_ours"] = all(type(new.model[i]) is mga_yolo_amd.MaskCBAM for i in (23, 25, 27))
            R["detect_from"] = new.model[28].f
            sd_new = new.state_dict()
            R["state_keys_equal"] = list(sd_init.keys()) == list(sd_new.keys())
            R["state_values_equal_same_seed"] = all(torch.equal(sd_init[k], sd_new[k]) for k in sd_init)
            new.load_state_dict(trained, strict=True)                   # a reference checkpoint into the patched model ...
            o_new = run_all(new)
            ref.load_state_dict(new.state_dict(), strict=True)          # ... and the patched model's state back into the reference
            R["cross_load_strict"] = all(torch.equal(new.state_dict()[k], trained_after[k]) or not trained_after[k].is_floating_point()
                                         or rel(new.state_dict()[k], trained_after[k]) < 1e-5 for k in trained_after)
            R["eval_forward_rel_diff"] = max_diff(o_new["eval"], o_ref["eval"])
            R["train_forward_rel_diff"] = max_diff(o_new["train"], o_ref["train"])
            R["seg_criterion"] = {"ref": o_ref["crit"], "new": o_new["crit"]}
            R["loss_ref"], R["loss_new"] = o_ref["loss"], o_new["loss"]
            gr, gn = o_ref["grads"], o_new["grads"]
            R["grad_keys_equal"] = sorted(gr) == sorted(gn)
            worst = max(((rel(gn[k], gr[k]), k) for k in gr if gr[k].abs().max() > 0), default=(0.0, ""))
            R["grad_worst_rel_diff"], R["grad_worst_key"] = worst
            R["block_grad_rel_diff"] = {k: rel(gn[k], gr[k]) for k in gr if k.split(".")[1:2] in (["23"], ["25"], ["27"])}
            # trainer's alpha collection (model/trainer.py:286-295): late import of the class + isinstance + .alpha
            from mga_yolo.nn.modules.masked_cbam import MaskCBAM as Late
            found = [float(m.alpha.detach()) for m in new.modules() if isinstance(m, Late)]
            R["alpha_found"] = found
            R["late_import_is_ours"] = Late is mga_yolo_amd.MaskCBAM
            # EMA deep-copies the model (U/utils/torch_utils.py:722)
            import copy
            cp = copy.deepcopy(new)
            R["deepcopy_ok"] = all(type(cp.model[i]) is mga_yolo_amd.MaskCBAM for i in (23, 25, 27))
            R["mask_heads"] = [f"{type(new.model[i]).__module__}.{type(new.model[i]).__name__}" for i in (22, 24, 26)]
            R["ref_head_class"] = f"{RefHead.__module__}.{RefHead.__name__}"
            report["scales"][scale] = R
            print(scale, json.dumps({k: v for k, v in R.items() if k not in ("layers", "block_grad_rel_diff", "patched_modules")}))
            mga_yolo_amd.uninstall()        # the next scale's reference build must see the reference classes again
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(OUT, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
