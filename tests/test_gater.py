"""ProbMaskGater.  Pinned by outputs of the REFERENCE module (oracle/gen_golden_gater.py ran it with a seed and
stored the input, the two uniform tensors it consumed, its output and its input gradient).
CPU: the module's host path reproduces the reference's draws and results from the same seed.
GPU: the HIP launch, fed the stored uniforms, equals the reference element-wise (value, hard decisions bit-exact, gradient)."""
import glob
import os

import numpy as np
import pytest
import torch

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "gater_*.npz")))


def _load(path):
    z = np.load(path)
    t = lambda k: torch.from_numpy(z[k].copy())
    return dict(p=t("p"), u1=t("u1"), u2=t("u2"), out=t("out"), gout=t("gout"), gp=t("gp"), mode=str(z["mode"]), tau=float(z["tau"]),
                p_min=float(z["p_min"]), threshold=float(z["threshold"]), seed=int(z["seed"]))


def test_goldens_exist():
    assert len(GOLD) >= 5


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_host_path_reproduces_the_reference(path):
    from mga_yolo_amd import ProbMaskGater
    c = _load(path)
    m = ProbMaskGater(mode=c["mode"], tau=c["tau"], p_min=c["p_min"], threshold=c["threshold"], seed=c["seed"]).train()
    x = c["p"].clone().requires_grad_(True)
    out = m(x)
    out.backward(c["gout"])
    assert out.shape == c["out"].shape
    assert torch.allclose(out.detach(), c["out"], rtol=1e-6, atol=1e-7)
    assert torch.allclose(x.grad, c["gp"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_device_launch_equals_the_reference(built_lib, path):
    from mga_yolo_amd import prob_mask_gate
    c = _load(path)
    p = c["p"] if c["p"].dim() == 4 else c["p"].unsqueeze(1)
    x = p.cuda().requires_grad_(True)
    out = prob_mask_gate(x, c["u1"].cuda(), c["u2"].cuda(), c["tau"], c["p_min"], c["threshold"], hard=c["mode"] == "hard_st")
    out.backward(c["gout"].cuda())
    if c["mode"] == "hard_st":
        m_ref = c["out"]
        flips = int((out.detach().cpu() != m_ref).sum())
        assert flips == 0, f"{flips} hard decisions differ"
    else:
        assert torch.allclose(out.detach().cpu(), c["out"], rtol=2e-6, atol=1e-7)
    gp = x.grad.cpu().reshape(c["gp"].shape)
    assert float((gp - c["gp"]).abs().max()) <= 1e-5 * float(c["gp"].abs().max()) + 1e-7


@pytest.mark.gpu
def test_device_module_path_runs_the_launch_and_is_seed_reproducible(built_lib):
    from mga_yolo_amd import ProbMaskGater
    p = torch.rand(4, 1, 20, 20).cuda().requires_grad_(True)
    outs = []
    for _ in range(2):
        m = ProbMaskGater(mode="gumbel", tau=0.8, seed=5).cuda().train()
        outs.append(m(p))
    assert torch.equal(outs[0], outs[1]) and outs[0].grad_fn is not None and "GaterFn" in type(outs[0].grad_fn).__name__
    outs[0].sum().backward()
    assert torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0
    m.eval()
    assert torch.equal(m(p), p.detach().clamp(0, 1))                           # eval: deterministic clamp, torch ops
    hard = ProbMaskGater(mode="hard_st", seed=1).cuda().train()(p)
    assert set(hard.detach().unique().tolist()) <= {0.0, 1.0}
