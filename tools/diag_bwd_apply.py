"""Diagnostics for k_bwd_apply / k_chan variants (GPU box): back-to-back launch time of single stages under different plans."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mga_yolo_amd import MaskCBAM, _lib
from mga_yolo_amd.plan import PyramidPlan

LV = {"all": [(64, 80, 80), (128, 40, 40), (256, 20, 20)], "p3": [(64, 80, 80)], "p4": [(128, 40, 40)], "p5": [(256, 20, 20)]}


def mk(levels, **kw):
    shapes, params, cfgs = [], [], []
    for C, H, W in levels:
        torch.manual_seed(0); m = MaskCBAM(C)
        shapes.append((32, C, H, W)); params.append(m.block_params()); cfgs.append(m.block_config())
    p = PyramidPlan(shapes, params, cfgs, **kw)
    g = torch.Generator().manual_seed(1)
    for l, s in enumerate(shapes):
        p.x[l].copy_(torch.randn(*s, generator=g)); p.gy[l].copy_(torch.randn(*s, generator=g))
        if p.mask[l] is not None:
            p.mask[l].copy_(torch.randn(s[0], 1, s[2], s[3], generator=g) - 2)
    p.forward(); p.backward(); torch.cuda.synchronize()
    return p


def t(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


B, Fw = _lib.BWD_STAGES, _lib.FWD_STAGES
for name, lv in LV.items():
    for kw in (dict(want_gmask=True, use_proj=True), dict(want_gmask=True, use_proj=False), dict(want_gmask=False)):
        p = mk(lv, **kw)
        r = dict(chan=t(lambda: p.forward(Fw["chan"])), apply=t(lambda: p.forward(Fw["apply"])),
                 bapply=t(lambda: p.backward(B["apply"])), bapply_fused=t(lambda: p.backward(B["apply"] | B["params"] | _lib.BWD_FUSE)),
                 params=t(lambda: p.backward(B["params"])), reduce1=t(lambda: p.backward(B["reduce1"])))
        print(f"{name:4s} {str(kw):44s} " + "  ".join(f"{k}={v:6.2f}" for k, v in r.items()), flush=True)
        del p
