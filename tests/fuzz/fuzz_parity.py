"""Randomised parity fuzz on the GPU box: many small random shapes / configurations of MaskCBAM and MaskECA through the HIP
path vs the oracles (1e-4 relative).  Exercises every launch-geometry branch (H*W odd / multiple of 4, C below / above the row
counts, B not a multiple of 8, generic conv sizes, no mask, raw-probability masks, tiny / empty masks, hidden 1..48).
    python tests/fuzz/fuzz_parity.py [n_cases] [seed]
"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import rel_err, synth
from oracle import maskcbam_oracle as O, maskeca_oracle as E
from mga_yolo_amd import functional as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    B = rng.choice([1, 2, 3, 5, 8, 9, 16])
    C = rng.choice([1, 2, 3, 7, 8, 16, 24, 48, 64, 96, 130, 256, 320])
    H, W = rng.randint(1, 28), rng.randint(1, 28)
    kind = rng.choice(["randn", "sparse", "mixed", "none", "all_negative", "tiny", "prob"])
    if kind == "mixed" and B < 2:
        kind = "randn"
    use_sig = kind != "prob"
    x, mask, gy = synth(B, C, H, W, seed=1000 + it, mask_kind=kind, x_kind=rng.choice(["randn", "relu", "quantized"]))
    block = rng.choice(["cbam", "cbam", "eca"])
    try:
        if block == "cbam":
            r = rng.choice([1, 2, 4, 16]); k = rng.choice([1, 3, 5, 7, 7, 9])
            p = O.Params.default_init(C, r=r, k=k, seed=it)
            with torch.no_grad():
                for t in (p.w1, p.b1, p.w2, p.b2, p.wsa):
                    t.add_(0.3 * torch.randn(t.shape))
                p.beta.fill_(rng.uniform(-1.5, 1.5))
            cfg = O.Config(use_sigmoid_mask=use_sig)
            y_o, c = O.forward(x, mask, p, cfg)
            g_o = O.backward(gy, x, mask, p, cfg, c)
            xd = x.cuda().requires_grad_(True)
            md = None if mask is None else mask.cuda().requires_grad_(True)
            ps = [t.cuda().requires_grad_(True) for t in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
            y = F.mask_cbam(xd, md, *ps, F.BlockConfig(hidden=p.w1.shape[0], k=k, use_sigmoid_mask=use_sig))
            y.backward(gy.cuda())
            got = dict(y=y, gx=xd.grad, gmask=None if md is None else md.grad, gw1=ps[0].grad, gb1=ps[1].grad, gw2=ps[2].grad,
                       gb2=ps[3].grad, gwsa=ps[4].grad, gbeta=ps[5].grad)
            want = dict(y=y_o, **g_o)
        else:
            p = E.EcaParams.default_init(C, seed=it)
            with torch.no_grad():
                p.w.add_(0.3 * torch.randn(p.w.shape)); p.beta.fill_(rng.uniform(-1.5, 1.5))
            cfg = E.EcaConfig(use_sigmoid_mask=use_sig)
            y_o, t = E.forward(x, mask, p, cfg)
            g_o = E.backward(gy, x, mask, p, cfg, t)
            xd = x.cuda().requires_grad_(True)
            md = None if mask is None else mask.cuda().requires_grad_(True)
            w, beta = p.w.cuda().requires_grad_(True), p.beta.cuda().requires_grad_(True)
            y = F.mask_eca(xd, md, w, beta, F.EcaConfig(k=p.w.shape[-1], use_sigmoid_mask=use_sig))
            y.backward(gy.cuda())
            got = dict(y=y, gx=xd.grad, gmask=None if md is None else md.grad, gw=w.grad, gbeta=beta.grad)
            want = dict(y=y_o, **g_o)
        # parameter gradients (and gmask) are long signed sums that may cancel to ~0: on top of 1e-4 relative allow the fp32
        # rounding of their terms, 1e-7 of |gy|.|x| (absolute)
        floor = 1e-7 * float(gy.norm() * x.norm())
        errs = {}
        for k_ in want:
            if want[k_] is None:
                continue
            wv = want[k_].double()
            dv = (got[k_].detach().double().cpu() - wv).abs().max()
            tol = 1e-4 * float(wv.abs().max()) + (floor if k_ not in ("y", "gx") else 0.0)
            errs[k_] = 1e-4 * float(dv) / max(tol, 1e-30)          # normalised so that the bar stays "< 1e-4"
        worst = max(errs.values()) if errs else 0.0
        if not worst < 1e-4:
            bad += 1
            print(f"FAIL case {it}: {block} B={B} C={C} H={H} W={W} mask={kind} -> {({k_: f'{v:.2e}' for k_, v in errs.items() if v >= 1e-4})}", flush=True)
            for k_, v in errs.items():
                if v >= 1e-4 and want[k_].numel() <= 16:
                    print(f"   {k_}: got {got[k_].detach().cpu().flatten().tolist()} want {want[k_].flatten().tolist()} floor {floor:.3e}", flush=True)
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print(f"ERROR case {it}: {block} B={B} C={C} H={H} W={W} mask={kind}: {type(ex).__name__}: {ex}", flush=True)
print(f"fuzz: {n - bad}/{n} cases within 1e-4")
sys.exit(1 if bad else 0)
