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
    ("hid192",      1, 768, 192,  4,  5, dict(randomize=25)),                           # x-scale P4/P5 width: hidden > 128 (four-tile GEMM template, its own launch)
    ("dc_offset",   2,  64,  16, 12, 12, dict(randomize=27, x_offset=50.0)),            # |mean| >> std per channel: variance by sum z^2 - n mean^2 would cancel
    ("w160_hid64",  1,  64,  64,  3, 160, dict(randomize=29)),                          # 1280-px P3 row width at m/l hidden width: the 3x3 kernels' LDS staging bound
]
BIG = [
    ("cfg2_p3", 32,  64, 16, 80, 80), ("cfg2_p4", 32, 128, 32, 40, 40), ("cfg2_p5", 32, 256, 64, 20, 20),
    ("cfg3_p3", 32, 128, 32, 80, 80), ("cfg3_p4", 32, 256, 64, 40, 40), ("cfg3_p5", 32, 512, 128, 20, 20),
    # BASELINE configs[4] (YOLOv8l + seg head, mixed 640 / 1280, bf16): the reference runs in fp32 on the bf16-rounded inputs;
    # configs[3]'s "x192" P3 at 1280 px (hidden 48: not a multiple of 16)
    ("cfg5_640_p3", 8, 256, 64, 80, 80, "bf16"), ("cfg5_640_p4", 8, 512, 128, 40, 40, "bf16"), ("cfg5_640_p5", 8, 512, 128, 20, 20, "bf16"),
    ("cfg5_1280_p3", 4, 256, 64, 160, 160, "bf16"), ("cfg4_p3_192", 8, 192, 48, 160, 160),
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


def data(B, C, H, W, seed=1234, x_kind="randn", x_offset=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    if x_kind == "zeros":
        x = torch.zeros(B, C, H, W)
    if x_offset:
        x = x + x_offset
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
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]      # --only=NAME[,NAME]: add cases / checksums, leave the committed ones alone
    only = [n for a in only for n in a.split(",")]
    for name, B, C, hid, H, W, kw in [c for c in CASES if not only or c[0] in only]:
        m = build(C, hid, **kw)
        x, gl = data(B, C, H, W, x_kind=kw.get("x_kind", "randn"), x_offset=kw.get("x_offset", 0.0))
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
    path = os.path.join(OUT, "head_checksums.json")
    sums = dict(json.load(open(path))["big"]) if only else {}
    for name, B, C, hid, H, W, *recipe in BIG:
        if only and (name in sums or name not in only):
            continue
        m = build(C, hid, eps=1e-3, momentum=0.03)
        x, gl = data(B, C, H, W)
        if recipe == ["bf16"]:
            x, gl = x.bfloat16().float(), gl.bfloat16().float()
        out, _ = run(m, x, gl)
        sums[name] = dict(shape=[B, C, hid, H, W], eps=1e-3, momentum=0.03, recipe=(recipe[0] if recipe else "fp32"),
                          **{k: checksum(v.float()) for k, v in out.items() if k != "num_batches_tracked"})
        print(f"big  {name:10s} logits.sum={sums[name]['logits']['sum']:.6f} |gx|={sums[name]['gx']['abs']:.6f}")
    with open(path, "w") as f:
        json.dump(dict(big=sums, torch=torch.__version__), f, indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
