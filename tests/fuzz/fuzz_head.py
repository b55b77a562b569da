"""Randomised parity fuzz of the mask head on the GPU box: MGAMaskHead (HIP path: mgahead_forward / mgahead_backward) vs its oracle on
random shapes -- channels / hidden widths that are not multiples of 16 or 4, odd H*W (scalar-lane path), tiny and wide images, hidden up
to 384 (several M blocks), train and eval mode, fp32 / fp16 / bf16 features, single and multi-level calls.
    python tests/fuzz/fuzz_head.py [n_cases] [seed]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import maskhead_oracle as HO
from mga_yolo_amd import MGAMaskHead, mask_head_pyramid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


for it in range(n):
    g = torch.Generator().manual_seed(7000 + it)
    B = rng.choice([1, 2, 3, 5, 8])
    C = rng.choice([3, 8, 20, 48, 64, 96, 128, 130, 192, 256, 320, 512])
    hid = rng.choice([1, 4, 8, 12, 16, 24, 32, 40, 64, 96, 128, 200, 256, 384])
    H, W = rng.choice([(1, 1), (1, 9), (2, 3), (5, 7), (8, 8), (9, 7), (12, 20), (17, 17), (20, 20), (16, 40), (33, 5), (40, 40),
                        (3, 160), (2, 333), (1, 500), (2, 257), (6, 100), (3, 96)])      # wide rows: k_head_out's pixels-per-thread escalation, tiny outputs per run
    if B * C * H * W > 6_000_000:
        B = 1
    dt = rng.choice([torch.float32, torch.float32, torch.float32, torch.float16, torch.bfloat16])
    training = rng.random() < 0.75
    desc = f"B={B} C={C} hid={hid} H={H} W={W} {dt} train={training}"
    try:
        torch.manual_seed(it)
        m = MGAMaskHead(C, hid)
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.3 * torch.randn(p_.shape, generator=g))
            bn = m.proj[1]
            bn.running_mean.add_(0.2 * torch.randn(hid, generator=g)); bn.running_var.mul_(0.5 + torch.rand(hid, generator=g))
        bn.eps, bn.momentum = rng.choice([(1e-5, 0.1), (1e-3, 0.03)])
        m.train(training)
        x = torch.randn(B, C, H, W, generator=g).to(dt)
        gl = torch.randn(B, 1, H, W, generator=g).to(dt)
        p = HO.HeadParams.from_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()}, eps=bn.eps, momentum=bn.momentum)
        lo, c = HO.forward(x.float(), p, training)
        go = HO.backward(gl.float(), x.float(), p, c, training)
        md = m.cuda()
        xd = x.cuda().requires_grad_(True)
        y = md(xd)
        y.backward(gl.cuda())
        tol = {torch.float32: 2e-4, torch.float16: 6e-3, torch.bfloat16: 4e-2}[dt]
        n_px = B * H * W
        if training and n_px < 4:
            tol = max(tol, 1e-1)          # batch statistics over 1-3 values: zhat = +-1, the BatchNorm backward cancels to eps-level terms and the
                                          # fp32 oracle is itself only good to a few percent of what is left (v_exp/v_rcp SiLU: 1e-6 in, 4e-2 out)
        errs = dict(logits=rel(y.float(), lo), gx=rel(xd.grad.float(), go["gx"]), gw1=rel(md.proj[0].weight.grad.reshape(hid, C), go["gw1"]),
                    ggamma=rel(md.proj[1].weight.grad, go["ggamma"]), gbeta=rel(md.proj[1].bias.grad, go["gbeta"]),
                    gwh=rel(md.head.weight.grad, go["gwh"]), gbh=rel(md.head.bias.grad, go["gbh"]),
                    rmean=rel(md.proj[1].running_mean, c.new_running_mean), rvar=rel(md.proj[1].running_var, c.new_running_var))
        if training and n_px < 4:
            # gx / dW1 through a BatchNorm over 2-3 values cancel to rounding noise in exact arithmetic (zhat = +-1): there is no signal to compare
            errs.pop("gx"); errs.pop("gw1")
        if n_px < 4 and dt != torch.float32:
            # one to three logits from half-precision features: the relative error of a single cancelling dot product (weights rounded to the
            # feature type for the MFMA) is not bounded by the tensor-scale tolerance -- nothing to compare at this size
            continue
        worst = max(errs, key=errs.get)
        if not errs[worst] < tol or any(v != v for v in errs.values()):
            bad += 1
            print(f"FAIL {it}: {desc}: {worst} {errs[worst]:.3e}  all={ {k: f'{v:.1e}' for k, v in errs.items()} }", flush=True)
    except ValueError as e:                                  # one value per channel in training mode: torch raises, and so does the HIP path
        if training and B * H * W == 1 and "more than 1 value per channel" in str(e):
            continue
        bad += 1
        print(f"ERROR {it}: {desc}: {type(e).__name__}: {e}", flush=True)
    except Exception as e:                                   # noqa: BLE001
        bad += 1
        print(f"ERROR {it}: {desc}: {type(e).__name__}: {e}", flush=True)
    if it % 50 == 49:
        print(f"  {it + 1} cases, {bad} bad", flush=True)
print(f"fuzz_head: {n} cases, {bad} bad")
sys.exit(1 if bad else 0)
