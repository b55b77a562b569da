"""MaskECA: oracle/maskeca_oracle.py pinned to outputs of the reference module (tests/golden/eca_*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from oracle import maskeca_oracle as E

ECA_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("eca_") and f.endswith(".npz"))


def load_eca(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    return dict(x=torch.from_numpy(z["x"]), gy=torch.from_numpy(z["gy"]),
                mask=torch.from_numpy(z["mask"]) if "mask" in z.files else None,
                params={k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")},
                out={k[len("out."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out.")},
                meta=json.loads(bytes(z["meta"]).decode()))


def _pc(d):
    p = E.EcaParams(d["params"]["conv1d.weight"], d["params"]["beta"])
    cfg = E.EcaConfig(use_sigmoid_mask=d["meta"]["use_sigmoid_mask"], tiny_thr=d["meta"]["tiny_thr"], eps=d["meta"]["eps"])
    return p, cfg


@pytest.mark.parametrize("name", ECA_CASES)
def test_explicit_matches_reference(name):
    d = load_eca(name)
    p, cfg = _pc(d)
    y, t = E.forward(d["x"], d["mask"], p, cfg)
    g = E.backward(d["gy"], d["x"], d["mask"], p, cfg, t)
    assert rel_err(y, d["out"]["y"]) < 2e-5
    for k in ("gx", "gmask", "gw", "gbeta"):
        if g[k] is None:
            assert "gmask" not in d["out"]
            continue
        assert g[k].shape == d["out"][k].shape, k
        assert rel_err(g[k], d["out"][k]) < 1e-4, k


@pytest.mark.parametrize("name", ECA_CASES)
def test_eager_form_matches_reference(name):
    d = load_eca(name)
    p, cfg = _pc(d)
    y, g = E.reference_form_step(d["x"], d["mask"], p, cfg, d["gy"])
    assert rel_err(y, d["out"]["y"]) < 1e-6
    for k, v in g.items():
        if v is not None:
            assert rel_err(v, d["out"][k]) < 1e-5, k


def test_kernel_size_rule_and_default_init():
    assert [E.eca_kernel_size(c) for c in (8, 64, 128, 256, 512, 1024)] == [3, 5, 5, 5, 5, 7]
    d = load_eca("eca_base")
    p = E.EcaParams.default_init(64)
    assert torch.equal(p.w, d["params"]["conv1d.weight"]) and p.w.shape == (1, 1, 5)
