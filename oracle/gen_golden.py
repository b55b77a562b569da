"""Generate the golden vectors that pin ``oracle/maskcbam_oracle.py`` (and through it the HIP kernels).

TEST INFRASTRUCTURE.  Run ONCE in the build container, where the upstream reference is mounted read-only:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python oracle/gen_golden.py

It imports the reference's own ``mga_yolo.nn.modules.masked_cbam.MaskCBAM`` (torch-only, imports cleanly
here -- SURVEY.md section 8c), runs forward + backward on seeded inputs and writes

* ``tests/golden/case_*.npz``      -- inputs, parameters, y and every gradient for small shapes
* ``tests/golden/checksums.json``  -- float64 checksums of y / gradients for full-size shapes
                                      (BASELINE.json configs 1 and 2) that are too large to commit

Only data is written -- no reference source text.  The reference never travels to the GPU box.
"""
import json
import os
import sys

import numpy as np
import torch

from mga_yolo.nn.modules.masked_cbam import MaskCBAM  # the reference itself
from mga_yolo.nn.modules.masked_eca import MaskECA    # next row

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_num_threads(1)  # deterministic summation order


def build(C, r=16, k=7, use_sigmoid_mask=True, seed=0, randomize=None, beta=None):
    torch.manual_seed(seed)
    m = MaskCBAM(C, r=r, spatial_k=k, use_sigmoid_mask=use_sigmoid_mask)
    if randomize is not None:  # "trained-like" parameters: every entry perturbed
        g = torch.Generator().manual_seed(randomize)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.5 * torch.randn(p_.shape, generator=g))
    if beta is not None:
        with torch.no_grad():
            m.beta.fill_(beta)
    return m


def run(m, x, mask, gy):
    x = x.clone().requires_grad_(True)
    mk = None if mask is None else mask.clone().requires_grad_(True)
    m.zero_grad()
    y = m(x if mk is None else [x, mk])
    y.backward(gy)
    out = dict(y=y.detach(), gx=x.grad)
    if mk is not None:
        out["gmask"] = mk.grad
    sd = m.state_dict()
    names = {"gw1": "cam_mlp.0.weight", "gb1": "cam_mlp.0.bias", "gw2": "cam_mlp.2.weight",
             "gb2": "cam_mlp.2.bias", "gwsa": "sam_conv.weight", "gbeta": "beta"}
    params = dict(m.named_parameters())
    for g_, n_ in names.items():
        out[g_] = params[n_].grad.detach().clone()
    return out, {k_: v.detach().clone() for k_, v in sd.items()}


def data(B, C, H, W, seed=1234, mask_kind="randn", mask3d=False, x_kind="randn"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    if x_kind == "quantized":      # many exact ties, for first-index tie-breaking
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
    elif mask_kind == "all_negative":     # nothing selected -> masked max falls back to GAP
        mask = -r.abs() - 0.1
    elif mask_kind == "tiny":             # sigmoid(-20) ~ 2e-9 -> use_mask = 0 -> masked avg falls back to GAP
        mask = torch.full(ms, -20.0)
    elif mask_kind == "sparse":           # vessel-like coverage
        mask = r - 2.0
    elif mask_kind == "prob":             # use_sigmoid_mask=False: mask already a probability
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
    elif mask_kind == "mixed":            # per-sample branches: normal / nothing selected / tiny
        mask = r.clone()
        mask[1] = -mask[1].abs() - 0.1
        if B > 2:
            mask[2] = -20.0
    else:
        raise ValueError(mask_kind)
    gy = torch.randn(B, C, H, W, generator=g)
    return x, mask, gy


CASES = [
    # name,            B,  C,  H,  W, module kwargs,                         data kwargs
    ("base",           2, 64, 20, 20, dict(),                                 dict()),
    ("nomask",         2, 32, 12, 12, dict(),                                 dict(mask_kind="none")),
    ("all_negative",   2, 16, 10, 10, dict(),                                 dict(mask_kind="all_negative")),
    ("tiny_mask",      2, 16, 10, 10, dict(),                                 dict(mask_kind="tiny")),
    ("mask3d",         1, 16,  8,  8, dict(),                                 dict(mask3d=True)),
    ("hidden1_nonsq",  2,  8,  9,  7, dict(),                                 dict()),
    ("odd17_c48",      1, 48, 17, 17, dict(),                                 dict()),
    ("c192",           1, 192, 8,  8, dict(),                                 dict()),
    ("ties",           2, 16,  8,  8, dict(randomize=7),                      dict(x_kind="quantized")),
    ("relu_sparse",    2, 32, 12, 16, dict(randomize=3),                      dict(x_kind="relu", mask_kind="sparse")),
    ("prob_mask",      2, 16,  8,  8, dict(use_sigmoid_mask=False),           dict(mask_kind="prob")),
    ("mixed_batch",    3, 32, 10, 10, dict(randomize=11),                     dict(mask_kind="mixed")),
    ("k3",             1, 16,  8,  8, dict(k=3, randomize=5),                 dict()),
    ("k5",             1, 16,  8,  8, dict(k=5, randomize=5),                 dict()),
    ("r4",             2, 32,  8,  8, dict(r=4, randomize=9),                 dict()),
    ("beta_pos",       2, 32,  8,  8, dict(randomize=13, beta=0.7),           dict()),
    ("beta_neg",       2, 32,  8,  8, dict(randomize=13, beta=-1.3),          dict()),
    ("stride_probe",   1, 64, 32, 32, dict(),                                 dict(x_kind="zeros", mask_kind="zeros")),
    ("p5_like",        2, 256, 5,  5, dict(randomize=17),                     dict()),
    ("w1_tail",        2, 16,  6, 10, dict(randomize=19),                     dict(mask_kind="sparse")),
    ("boundary_logits", 2, 24, 12, 20, dict(randomize=23),                    dict(mask_kind="boundary")),
]

# full-size shapes: only checksums are kept (x is 3-52 MB)
BIG = [
    # name,                 B,  C,  H,  W, mask_kind, recipe
    ("survey_A1",           2, 64, 80, 80, "randn", "ysum"),       # SURVEY.md appendix A1 / section 8c case (1)
    ("survey_nomask",       2, 64, 80, 80, "none", "ysum"),        # case (2)
    ("cfg1_p3",             2, 64, 80, 80, "randn", "gy"),
    ("cfg1_p4",             2, 128, 40, 40, "randn", "gy"),
    ("cfg1_p5",             2, 256, 20, 20, "randn", "gy"),
    ("cfg2_p3",            32, 64, 80, 80, "sparse", "gy"),
    ("cfg2_p4",            32, 128, 40, 40, "sparse", "gy"),
    ("cfg2_p5",            32, 256, 20, 20, "sparse", "gy"),
    # BASELINE.json configs[2] (YOLOv8s, 32 images per GPU), configs[3] (YOLOv8m @1280, 8 per GPU), configs[4] (YOLOv8l, mixed 640/1280,
    # bf16: the reference runs in fp32 on the bf16-rounded inputs), per-GPU batch.  `--big-only` adds them without rewriting the cases.
    ("cfg3_p3",            32, 128, 80, 80, "sparse", "gy"),
    ("cfg3_p4",            32, 256, 40, 40, "sparse", "gy"),
    ("cfg3_p5",            32, 512, 20, 20, "sparse", "gy"),
    ("cfg4_p3",             8, 256, 160, 160, "sparse", "gy"),
    ("cfg4_p4",             8, 512, 80, 80, "sparse", "gy"),
    ("cfg4_p5",             8, 512, 40, 40, "sparse", "gy"),
    ("cfg5_640_p3",         8, 256, 80, 80, "sparse", "gy_bf16"),
    ("cfg5_640_p4",         8, 512, 40, 40, "sparse", "gy_bf16"),
    ("cfg5_640_p5",         8, 512, 20, 20, "sparse", "gy_bf16"),
    ("cfg5_1280_p3",        4, 256, 160, 160, "sparse", "gy_bf16"),
    ("cfg5_1280_p4",        4, 512, 80, 80, "sparse", "gy_bf16"),
    ("cfg5_1280_p5",        4, 512, 40, 40, "sparse", "gy_bf16"),
]


ECA_CASES = [
    # name,          B,  C,   H,  W, module kwargs,                     data kwargs
    ("eca_base",      2, 64,  20, 20, dict(),                            dict()),
    ("eca_nomask",    2, 32,  12, 12, dict(),                            dict(mask_kind="none")),
    ("eca_tiny",      2, 16,  10, 10, dict(),                            dict(mask_kind="tiny")),
    ("eca_mixed",     3, 32,  10, 10, dict(randomize=11),                dict(mask_kind="mixed")),
    ("eca_mask3d",    1, 16,   8,  8, dict(randomize=2),                 dict(mask3d=True)),
    ("eca_odd",       2, 48,  17, 17, dict(randomize=4),                 dict(mask_kind="sparse")),
    ("eca_c256",      2, 256,  5,  5, dict(randomize=6, beta=0.8),       dict()),
    ("eca_c1024",     1, 1024, 2,  2, dict(randomize=8, beta=-0.7),      dict()),
    ("eca_prob",      2, 16,   8,  8, dict(use_sigmoid_mask=False),      dict(mask_kind="prob")),
    ("eca_c8_nonsq",  2, 8,    9,  7, dict(randomize=3),                 dict()),
    ("eca_kmin7",     1, 32,   6,  6, dict(k_min=7, randomize=9),        dict()),
]


def build_eca(C, use_sigmoid_mask=True, seed=0, randomize=None, beta=None, k_min=3):
    torch.manual_seed(seed)
    m = MaskECA(C, use_sigmoid_mask=use_sigmoid_mask, k_min=k_min)
    if randomize is not None:
        g = torch.Generator().manual_seed(randomize)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.5 * torch.randn(p_.shape, generator=g))
    if beta is not None:
        with torch.no_grad():
            m.beta.fill_(beta)
    return m


def run_eca(m, x, mask, gy):
    x = x.clone().requires_grad_(True)
    mk = None if mask is None else mask.clone().requires_grad_(True)
    m.zero_grad()
    y = m(x if mk is None else [x, mk])
    y.backward(gy)
    out = dict(y=y.detach(), gx=x.grad, gw=m.conv1d.weight.grad.detach().clone(), gbeta=m.beta.grad.detach().clone())
    if mk is not None:
        out["gmask"] = mk.grad
    return out, {k_: v.detach().clone() for k_, v in m.state_dict().items()}


def checksum(tn):
    t = tn.double().reshape(-1)
    n = t.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37)   # position-sensitive weight
    return dict(sum=float(t.sum()), abs=float(t.abs().sum()), wsum=float((t * w).sum()), n=n,
                first=float(t[0]), last=float(t[-1]))


def main():
    os.makedirs(OUT, exist_ok=True)
    index = {}
    big_only = "--big-only" in sys.argv
    path = os.path.join(OUT, "checksums.json")
    old = json.load(open(path)) if (big_only and os.path.exists(path)) else {}
    if big_only:
        torch.set_num_threads(8)       # checksums only: 1e-4-level comparisons do not depend on the summation order
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]      # --only=NAME: add a case, leave the committed ones alone
    if only:
        old = json.load(open(path))
        index = dict(old.get("cases", {}))
    for name, B, C, H, W, mk, dk in ([] if big_only else [c for c in CASES if not only or c[0] in only]):
        m = build(C, **mk)
        x, mask, gy = data(B, C, H, W, **dk)
        out, sd = run(m, x, mask, gy)
        arrays = dict(x=x.numpy(), gy=gy.numpy())
        if mask is not None:
            arrays["mask"] = mask.numpy()
        for k_, v in sd.items():
            arrays["param." + k_] = v.numpy()
        for k_, v in out.items():
            arrays["out." + k_] = v.numpy()
        meta = dict(r=mk.get("r", 16), k=m.k, use_sigmoid_mask=bool(m.use_sigmoid_mask), tiny_thr=m.tiny_thr, eps=m.eps)
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), **arrays)
        index[name] = dict(shape=[B, C, H, W], **meta, ysum=float(out["y"].double().sum()))
        print(f"case {name:16s} y.sum={index[name]['ysum']:.6f}")

    for name, B, C, H, W, mk, dk in ([] if (big_only or only) else ECA_CASES):
        m = build_eca(C, **mk)
        x, mask, gy = data(B, C, H, W, **dk)
        out, sd = run_eca(m, x, mask, gy)
        arrays = dict(x=x.numpy(), gy=gy.numpy())
        if mask is not None:
            arrays["mask"] = mask.numpy()
        for k_, v in sd.items():
            arrays["param." + k_] = v.numpy()
        for k_, v in out.items():
            arrays["out." + k_] = v.numpy()
        meta = dict(k=int(m.conv1d.weight.shape[-1]), use_sigmoid_mask=bool(m.cfg.use_sigmoid_mask),
                    tiny_thr=m.cfg.tiny_mask_threshold, eps=m.cfg.eps, k_min=mk.get("k_min", 3))
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **arrays)
        index[name] = dict(shape=[B, C, H, W], **meta, ysum=float(out["y"].double().sum()))
        print(f"eca  {name:16s} k={meta['k']} y.sum={index[name]['ysum']:.6f}")

    sums = dict(old.get("big", {}))
    for name, B, C, H, W, mkind, recipe in BIG:
        if (big_only or only) and name in sums:
            continue
        m = build(C)
        x, mask, gy = data(B, C, H, W, mask_kind=mkind)
        if recipe == "ysum":
            gy = torch.ones_like(x)
        if recipe == "gy_bf16":
            x, gy = x.bfloat16().float(), gy.bfloat16().float()
        out, _ = run(m, x, mask, gy)
        sums[name] = dict(shape=[B, C, H, W], mask_kind=mkind, recipe=recipe,
                          **{k_: checksum(v) for k_, v in out.items()})
        print(f"big  {name:16s} y.sum={sums[name]['y']['sum']:.6f} |gx|={sums[name]['gx']['abs']:.6f}")
    with open(path, "w") as f:
        json.dump(dict(cases=old.get("cases", index) if big_only else index, big=sums,
                       torch=old.get("torch", torch.__version__) if (big_only or only) else torch.__version__),
                  f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
