"""Generate the golden vectors that pin ``oracle/maskhead_oracle.py`` and the HIP mask-head path.

TEST INFRASTRUCTURE.  Run ONCE in the build container, where the upstream reference is mounted read-only:

    PYTHONDONTWRITEBYTECODE=1 YOLO_CONFIG_DIR=/tmp/yolo_cfg python oracle/gen_golden_head.py

The reference's ``mga_yolo/nn/modules/segmentation.py`` imports Ultralytics' LOGGER at module scope (:30), whose package imports
``cv2`` (absent from this image): the in-process stand-in of SURVEY appendix A2 lets the reference's own ``MGAMaskHead`` be imported
and run (nothing of the stand-in is executed on this path).  Written:

* ``tests/golden/head_*.npz``  -- input, every state_dict entry before the call, logits, BatchNorm running statistics after the
  call, and every gradient (train and eval mode);
* ``tests/golden/head_checksums.json`` -- float64 checksums for the BASELINE config-2 / config-3 head shapes (too large to commit).

Only data is written -- no reference source text.
"""
import importlib.metadata as md
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

cv2 = MagicMock(name="cv2"); cv2.__version__ = "4.10.0"; cv2.__spec__ = None
sys.modules["cv2"] = cv2
_real_version = md.version
md.version = lambda n: "0.25.0" if n == "torchvision" else _real_version(n)
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

import mga_yolo  # noqa: E402,F401
from mga_yolo.nn.modules.segmentation import MGAMaskHead  # noqa: E402  (the reference itself)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_num_threads(1)

# name, B, C, hid, H, W, kwargs
CASES = [
    ("p3_like",     2,  64,  16, 12, 12, dict()),
    ("p4_like",     2, 128,  32, 10, 10, dict(randomize=3)),
    ("p5_like",     2, 256,  64,  6,  6, dict(randomize=5)),
    ("ultra_bn",    3,  64,  16,  8, 16, dict(randomize=7, eps=1e-3, momentum=0.03)),    # what initialize_weights sets (U/utils/torch_utils.py:570-572)
    ("hid8_odd",    1,  48,   8, 17, 17, dict(randomize=9)),                            # hidden not a multiple of 16, H*W odd (544-px inputs)
    ("hid24_nonsq", 2,  96,  24,  9,  7, dict(randomize=11)),
    ("b1_c20",      1,  20,  16,  5,  4, dict(randomize=13)),                           # channels not a multiple of 4... of 16
    ("eval",        2,  64,  16, 12, 12, dict(randomize=15, training=False)),
    ("eval_ultra",  2, 128,  32,  8,  8, dict(randomize=17, training=False, eps=1e-3, momentum=0.03)),
    ("single_px",   4,  32,  16,  1,  1, dict(randomize=19)),
    ("wide",        1,  64,  16,  3, 40, dict(randomize=21)),
    ("hid128",      1, 512, 128,  4,  4, dict(randomize=23)),
    ("stride_probe", 1, 64,  16, 32, 32, dict(x_kind="zeros", training=False)),          # parse_model's eval-mode zero-image probe (U/nn/tasks.py:413-429)
]
BIG = [
    ("cfg2_p3", 32,  64, 16, 80, 80), ("cfg2_p4", 32, 128, 32, 40, 40), ("cfg2_p5", 32, 256, 64, 20, 20),
    ("cfg3_p3", 32, 128, 32, 80, 80), ("cfg3_p4", 32, 256, 64, 40, 40), ("cfg3_p5", 32, 512, 128, 20, 20),
]


def build(C, hid, seed=0, randomize=None, eps=None, momentum=None, training=True, **_):
    torch.manual_seed(seed)
    m = MGAMaskHead(C, hid)
    if randomize is not None:                       # "trained-like": every parameter and running statistic perturbed
        g = torch.Generator().manual_seed(randomize)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.3 * torch.randn(p_.shape, generator=g))
            bn = m.proj[1]
            bn.running_mean.add_(0.2 * torch.randn(bn.running_mean.shape, generator=g))
            bn.running_var.mul_(0.5 + torch.rand(bn.running_var.shape, generator=g))
            bn.num_batches_tracked.fill_(5)
    if eps is not None:
        m.proj[1].eps = eps
    if momentum is not None:
        m.proj[1].momentum = momentum
    m.train(training)
    return m


def data(B, C, H, W, seed=1234, x_kind="randn"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    if x_kind == "zeros":
        x = torch.zeros(B, C, H, W)
    gl = torch.randn(B, 1, H, W, generator=g)
    return x, gl


def run(m, x, gl):
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    xl = x.clone().requires_grad_(True)
    m.zero_grad()
    y = m(xl)
    y.backward(gl)
    after = {k: v.detach().clone() for k, v in m.state_dict().items()}
    params = dict(m.named_parameters())
    out = dict(logits=y.detach(), gx=xl.grad, gw1=params["proj.0.weight"].grad, ggamma=params["proj.1.weight"].grad,
               gbeta=params["proj.1.bias"].grad, gwh=params["head.weight"].grad, gbh=params["head.bias"].grad,
               running_mean=after["proj.1.running_mean"], running_var=after["proj.1.running_var"],
               num_batches_tracked=after["proj.1.num_batches_tracked"])
    return out, before


def checksum(tn):
    t = tn.double().reshape(-1)
    n = t.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37)
    return dict(sum=float(t.sum()), abs=float(t.abs().sum()), wsum=float((t * w).sum()), n=n, first=float(t[0]), last=float(t[-1]))


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, B, C, hid, H, W, kw in CASES:
        m = build(C, hid, **kw)
        x, gl = data(B, C, H, W, x_kind=kw.get("x_kind", "randn"))
        out, before = run(m, x, gl)
        arrays = dict(x=x.numpy(), g_logits=gl.numpy())
        for k, v in before.items():
            arrays["param." + k] = v.numpy()
        for k, v in out.items():
            arrays["out." + k] = v.numpy()
        bn = m.proj[1]
        meta = dict(training=bool(m.training), eps=bn.eps, momentum=bn.momentum, hidden=hid)
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, f"head_{name}.npz"), **arrays)
        print(f"head {name:14s} logits.sum={float(out['logits'].double().sum()):.6f} |gx|={float(out['gx'].abs().sum()):.6f}")
    torch.set_num_threads(8)
    sums = {}
    for name, B, C, hid, H, W in BIG:
        m = build(C, hid, eps=1e-3, momentum=0.03)
        x, gl = data(B, C, H, W)
        out, _ = run(m, x, gl)
        sums[name] = dict(shape=[B, C, hid, H, W], eps=1e-3, momentum=0.03, **{k: checksum(v.float()) for k, v in out.items() if k != "num_batches_tracked"})
        print(f"big  {name:10s} logits.sum={sums[name]['logits']['sum']:.6f} |gx|={sums[name]['gx']['abs']:.6f}")
    with open(os.path.join(OUT, "head_checksums.json"), "w") as f:
        json.dump(dict(big=sums, torch=torch.__version__), f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
