"""Shared fixtures.  ``-m gpu`` tests need an MI355X; everything else runs on CPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_case_names():
    return sorted(f[len("case_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("case_") and f.endswith(".npz"))


def load_golden(name):
    """-> dict(x, mask|None, gy, params{state_dict name: tensor}, out{y,gx,gmask,...}, meta)."""
    z = np.load(os.path.join(GOLDEN, f"case_{name}.npz"), allow_pickle=False)
    d = dict(x=torch.from_numpy(z["x"]), gy=torch.from_numpy(z["gy"]),
             mask=torch.from_numpy(z["mask"]) if "mask" in z.files else None,
             params={k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")},
             out={k[len("out."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out.")},
             meta=json.loads(bytes(z["meta"]).decode()))
    return d


def load_checksums():
    with open(os.path.join(GOLDEN, "checksums.json")) as f:
        return json.load(f)


def synth(B, C, H, W, seed=1234, mask_kind="randn", x_kind="randn", mask3d=False):
    """The seeded input recipe shared with oracle/gen_golden.py (same generator call order)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    if x_kind == "quantized":
        x = torch.round(x * 2.0) / 2.0
    elif x_kind == "relu":
        x = torch.relu(x)
    elif x_kind == "zeros":
        x = torch.zeros(B, C, H, W)
    ms = (B, H, W) if mask3d else (B, 1, H, W)
    r = torch.randn(ms, generator=g)
    if mask_kind == "randn":
        mask = r
    elif mask_kind == "none":
        mask = None
    elif mask_kind == "all_negative":
        mask = -r.abs() - 0.1
    elif mask_kind == "tiny":
        mask = torch.full(ms, -20.0)
    elif mask_kind == "sparse":
        mask = r - 2.0
    elif mask_kind == "prob":
        mask = torch.rand(ms, generator=g)
    elif mask_kind == "zeros":
        mask = torch.zeros(ms)
    elif mask_kind == "boundary":          # logits within a few ulp of the selector's threshold, scattered over a random field; the
        mask = r.clone()                   # feature is largest exactly there, so a pixel selected (or not) by mistake moves max / arg-max
        t = 1.5 * 2.0 ** -24               # sigmoid(m) > 0.5 in fp32 <=> m > 1.5 * 2^-24
        vals = [0.0, -0.0, t, -t, 2.0 ** -22, -2.0 ** -22, 2.0 ** -23, 2.0 ** -24, 2.0 ** -25, 2.0 ** -26, -2.0 ** -26, 1e-40, -1e-40]
        vals += [float(torch.nextafter(torch.tensor(t), torch.tensor(1.0))), float(torch.nextafter(torch.tensor(t), torch.tensor(0.0))),
                 float(torch.nextafter(torch.tensor(2.0 ** -23), torch.tensor(1.0))), 3.0 * 2.0 ** -25, 5.0 * 2.0 ** -26, 1e-7, 8.9e-8, 9.0e-8]
        flat = mask.view(-1)
        pos = torch.randperm(flat.numel(), generator=g)[: 6 * len(vals)]
        flat[pos] = torch.tensor(vals, dtype=torch.float32).repeat(6)
        xb = x.view(B, C, -1)
        hw = xb.shape[-1]
        for q, pp in enumerate(pos.tolist()):          # channel (q mod C) peaks at this pixel
            xb[pp // hw, q % C, pp % hw] = 6.0 + 0.01 * q
    elif mask_kind == "mixed":
        mask = r.clone()
        mask[1] = -mask[1].abs() - 0.1
        if B > 2:
            mask[2] = -20.0
    else:
        raise ValueError(mask_kind)
    gy = torch.randn(B, C, H, W, generator=g)
    return x, mask, gy


def checksum(tn):
    t = tn.detach().double().reshape(-1).cpu()
    n = t.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37)
    return dict(sum=float(t.sum()), abs=float(t.abs().sum()), wsum=float((t * w).sum()), n=n,
                first=float(t[0]), last=float(t[-1]))


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): relative to the tensor's scale (1e-4 relative fp32)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_err(a, b, floor=1e-3):
    """ELEMENT-wise relative error with an absolute floor: max_i |a_i - b_i| / max(|b_i|, floor * max|b|).  rel_err above is relative
    to the tensor's scale, which leaves small-magnitude elements (e.g. dL/dmask of nearly saturated pixels) unconstrained in relative
    terms; this bounds every element whose magnitude is at least `floor` of the scale."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.abs().clamp_min(floor * float(b.abs().max().clamp_min(1e-30)))
    return float(((a - b).abs() / den).max())


@pytest.fixture(scope="session")
def checksums():
    return load_checksums()


@pytest.fixture(scope="session")
def built_lib():
    """libmgacbam.so, (re)built in-tree if sources are newer (hipcc cross-compiles gfx950 without a GPU)."""
    from mga_yolo_amd import build as B
    try:
        return B.build()
    except RuntimeError as e:
        if os.path.exists(B.LIB):
            return B.LIB
        pytest.skip(f"cannot build libmgacbam.so here: {e}")
