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


# ---------------------------------------------------------------- module mirror on the host (CPU) -------------------------
@pytest.mark.parametrize("name", ECA_CASES)
def test_host_module_matches_reference_golden(name):
    from mga_yolo_amd import MaskECA
    d = load_eca(name)
    C = d["x"].shape[1]
    m = MaskECA(C, use_sigmoid_mask=d["meta"]["use_sigmoid_mask"], k_min=d["meta"]["k_min"])
    m.load_state_dict(d["params"])
    x = d["x"].clone().requires_grad_(True)
    mk = None if d["mask"] is None else d["mask"].clone().requires_grad_(True)
    y = m(x if mk is None else [x, mk])
    y.backward(d["gy"])
    assert rel_err(y, d["out"]["y"]) < 1e-5 and rel_err(x.grad, d["out"]["gx"]) < 1e-4
    assert rel_err(m.conv1d.weight.grad, d["out"]["gw"]) < 1e-4 and rel_err(m.beta.grad, d["out"]["gbeta"]) < 1e-4
    if mk is not None:
        assert rel_err(mk.grad, d["out"]["gmask"]) < 1e-4


def test_module_contract():
    from mga_yolo_amd import MaskECA
    torch.manual_seed(0)
    m = MaskECA(64)
    d = load_eca("eca_base")
    assert list(m.state_dict()) == ["beta", "conv1d.weight"] and torch.equal(m.conv1d.weight, d["params"]["conv1d.weight"])
    assert m.scale_name == "C64" and MaskECA(256).scale_name == "P3" and abs(float(m.alpha) - 0.6931472) < 1e-6
    assert "C=64" in m.extra_repr()
    m(torch.zeros(1, 128, 4, 4))                              # runtime channel change rebuilds the conv, as the reference
    assert m.cfg.channels == 128 and m.conv1d.weight.shape[-1] == 5


# ---------------------------------------------------------------- HIP path (-m gpu) -----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ECA_CASES)
def test_gpu_matches_reference_golden(name):
    from mga_yolo_amd import EcaConfig, mask_eca
    d = load_eca(name)
    cfg = EcaConfig(k=d["meta"]["k"], use_sigmoid_mask=d["meta"]["use_sigmoid_mask"], tiny_thr=d["meta"]["tiny_thr"], eps=d["meta"]["eps"])
    x = d["x"].cuda().requires_grad_(True)
    mk = None if d["mask"] is None else d["mask"].cuda().requires_grad_(True)
    w = d["params"]["conv1d.weight"].cuda().requires_grad_(True)
    beta = d["params"]["beta"].cuda().requires_grad_(True)
    y = mask_eca(x, mk, w, beta, cfg)
    y.backward(d["gy"].cuda())
    rep = []
    for nm, got, want in (("y", y, d["out"]["y"]), ("gx", x.grad, d["out"]["gx"]), ("gw", w.grad, d["out"]["gw"]),
                          ("gbeta", beta.grad, d["out"]["gbeta"])) + ((("gmask", mk.grad, d["out"]["gmask"]),) if mk is not None else ()):
        assert got.shape == want.shape, nm
        e = rel_err(got, want)
        if not e < 1e-4:
            rep.append(f"{nm} {e:.3e}")
    assert not rep, f"{name}: " + "; ".join(rep)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,mask_kind", [((32, 64, 80, 80), "sparse"), ((32, 256, 20, 20), "randn"), ((3, 48, 17, 17), "mixed"),
                                             ((2, 512, 40, 40), "mixed"), ((1, 1024, 10, 10), "randn"), ((5, 8, 3, 5), "randn")])
def test_gpu_vs_oracle_live(shape, mask_kind):
    from conftest import synth
    from mga_yolo_amd import MaskECA
    B, C, H, W = shape
    if mask_kind == "mixed" and B < 2:
        mask_kind = "randn"
    x, mask, gy = synth(B, C, H, W, seed=31, mask_kind=mask_kind)
    torch.manual_seed(1)
    m = MaskECA(C)
    with torch.no_grad():
        m.beta.fill_(0.4)
    p = E.EcaParams(m.conv1d.weight.detach().clone(), m.beta.detach().clone())
    y_o, t = E.forward(x, mask, p)
    g_o = E.backward(gy, x, mask, p, E.EcaConfig(), t)
    m.cuda()
    xd, md = x.cuda().requires_grad_(True), mask.cuda().requires_grad_(True)
    y = m([xd, md])
    y.backward(gy.cuda())
    assert rel_err(y, y_o) < 1e-4 and rel_err(xd.grad, g_o["gx"]) < 1e-4 and rel_err(md.grad, g_o["gmask"]) < 1e-4
    assert rel_err(m.conv1d.weight.grad, g_o["gw"]) < 1e-4 and rel_err(m.beta.grad, g_o["gbeta"]) < 1e-4
    # no mask, and a pyramid call over two levels
    y2 = m(x.cuda())
    assert rel_err(y2, E.forward(x, None, p)[0]) < 1e-4
