"""Randomised parity fuzz of the smaller rows on the GPU box: segmentation loss (both modes, random level counts / sizes / target
resolutions / weights / dtypes) vs its oracle, and the ProbMaskGater launch vs the module's host math on the same uniforms.
    python tests/fuzz/fuzz_rows.py [n_cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import segloss_oracle as SO
from mga_yolo_amd import ProbMaskGater, SegLossConfig, SegmentationLoss, prob_mask_gate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    g = torch.Generator().manual_seed(9000 + it)
    try:
        if it % 2 == 0:                                                      # ---- segmentation loss
            B = rng.choice([1, 2, 3, 7, 16])
            keys = rng.sample(["p3", "p4", "p5"], rng.randint(1, 3))
            dt = rng.choice([torch.float32, torch.float32, torch.float16, torch.bfloat16])
            preds, tg = {}, []
            for i, k in enumerate(("p3", "p4", "p5")):
                H, W = rng.randint(1, 40), rng.randint(1, 40)
                if k in keys:
                    preds[k] = (torch.randn(B, 1, H, W, generator=g) * rng.choice([0.5, 2.0, 8.0])).to(dt)
                th, tw = (H, W) if rng.random() < 0.5 else (rng.randint(1, 90), rng.randint(1, 90))
                t = (torch.rand(B, 1, th, tw, generator=g) > rng.choice([0.5, 0.9, 0.99])).float()
                tg.append(t.squeeze(1) if rng.random() < 0.3 else t)
            kw = dict(bce_weight=rng.uniform(0.2, 2), dice_weight=rng.uniform(0.2, 2), smooth=rng.choice([1.0, 0.1, 5.0]),
                      scale_weights=tuple(rng.uniform(0.2, 2) for _ in range(3)), loss_lambda=rng.uniform(0.3, 2),
                      use_unified_focal=rng.random() < 0.4, ufl_lambda=rng.uniform(0.1, 0.9), ufl_delta=rng.uniform(0.2, 0.8),
                      ufl_gamma=rng.uniform(0.2, 0.9))
            po = {k: v.float().clone().requires_grad_(True) for k, v in preds.items()}
            to, lo = SO.forward(po, tg, SO.SegLossConfig(**kw))
            to.backward()
            pd = {k: v.cuda().requires_grad_(True) for k, v in preds.items()}
            td, ld = SegmentationLoss(SegLossConfig(**kw))(pd, [t.cuda() for t in tg])
            td.backward()
            tol = {torch.float32: 1e-4, torch.float16: 6e-3, torch.bfloat16: 4e-2}[dt]     # half: the gradient is rounded to the logits' dtype
            errs = {k: abs(ld[k] - lo[k]) / max(1.0, abs(lo[k])) for k in lo}
            for k in preds:
                w = po[k].grad
                errs["g_" + k] = float((pd[k].grad.float().cpu() - w).abs().max()) / (float(w.abs().max()) + 1e-12) * (1e-4 / tol)
            worst = max(errs.values())
            if not worst < 1e-4:
                bad += 1
                print(f"FAIL case {it}: segloss {dt} B={B} keys={keys} ufl={kw['use_unified_focal']} -> "
                      f"{({k: f'{v:.2e}' for k, v in errs.items() if v >= 1e-4})}", flush=True)
        else:                                                                # ---- ProbMaskGater launch
            shape = (rng.randint(1, 6), 1, rng.randint(1, 50), rng.randint(1, 50))
            p = torch.rand(shape, generator=g) * 1.8 - 0.4
            u1, u2 = torch.rand(shape, generator=g), torch.rand(shape, generator=g)
            tau, pmin, thr, hard = rng.choice([0.3, 1.0, 2.5]), rng.choice([0.0, 0.0, 0.15]), rng.uniform(0.2, 0.8), rng.random() < 0.5
            gout = torch.randn(shape, generator=g)
            x = p.clone().requires_grad_(True)                               # host math on the same uniforms
            q = x.clamp(0.0, 1.0)
            if pmin > 0:
                q = q.clamp_min(pmin)
            lo_, hi_ = 1e-6, 1 - 1e-6
            noise = torch.log(-torch.log(u2.clamp(lo_, hi_))) - torch.log(-torch.log(u1.clamp(lo_, hi_)))
            qq = q.clamp(lo_, hi_)
            soft = torch.sigmoid((torch.log(qq) - torch.log1p(-qq) + noise) / tau)
            ref = (soft > thr).float() + (soft - soft.detach()) if hard else soft
            ref.backward(gout)
            xd = p.cuda().requires_grad_(True)
            out = prob_mask_gate(xd, u1.cuda(), u2.cuda(), tau, pmin, thr, hard)
            out.backward(gout.cuda())
            near = (soft.detach() - thr).abs() < 1e-6                        # decisions within rounding of the threshold may flip
            ok_v = bool(((out.detach().cpu() - ref.detach()).abs() <= 2e-6 + 1.0 * near.float()).all())
            gw = x.grad
            ok_g = float((xd.grad.cpu() - gw).abs().max()) <= 1e-4 * float(gw.abs().max()) + 1e-7
            if not (ok_v and ok_g):
                bad += 1
                print(f"FAIL case {it}: gater shape={shape} tau={tau} pmin={pmin} hard={hard} value_ok={ok_v} grad_ok={ok_g}", flush=True)
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print(f"ERROR case {it}: {type(ex).__name__}: {ex}", flush=True)
print(f"rows fuzz: {n - bad}/{n} cases within tolerance")
sys.exit(1 if bad else 0)
