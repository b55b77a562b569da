"""ProbMaskGater measurement on an MI355X: forward + backward of the Gumbel gate on the P3/P4/P5 masks of BASELINE
configs[1] (32 x 1 x {80,40,20}^2), module path (2 torch.rand launches + 1 HIP launch forward, 1 backward) against the same
module math as torch elementwise ops (what the reference launches on a GPU), eager launches both; prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mga_yolo_amd import ProbMaskGater

B, SIZES = 32, [(80, 80), (40, 40), (20, 20)]
ps = [torch.rand(B, 1, h, w, device="cuda").requires_grad_(True) for h, w in SIZES]
gs = [torch.randn(B, 1, h, w, device="cuda") for h, w in SIZES]
gate = ProbMaskGater(mode="gumbel", tau=1.0).cuda().train()


def hip_step():
    for p, g in zip(ps, gs):
        p.grad = None
        gate(p).backward(g)


def torch_step():                      # the module's own host-path math (= the reference's op sequence), on device tensors
    for p, g in zip(ps, gs):
        p.grad = None
        q = p.float().clamp(0.0, 1.0)
        gate._soft_sample(q).backward(g)


def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(json.dumps(dict(block="ProbMaskGater(gumbel)", unit="us per fwd+bwd over P3+P4+P5 masks", hip_module_path=round(timed(hip_step), 1),
                      torch_elementwise_ops=round(timed(torch_step), 1), elements=sum(B * h * w for h, w in SIZES))))
