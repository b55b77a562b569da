"""How long the drop-in path takes when driven like the reference trainer drives it: three nn.Module calls + autograd backward,
eager launches from Python (no hipGraph), vs the graph-replayed PyramidPlan step bench.py reports."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mga_yolo_amd import MaskCBAM

LV = [(64, 80, 80), (128, 40, 40), (256, 20, 20)]
B = 32
mods, xs, ms, gys = [], [], [], []
for C, H, W in LV:
    torch.manual_seed(0)
    mods.append(MaskCBAM(C).cuda())
    xs.append(torch.randn(B, C, H, W, device="cuda", requires_grad=True))
    ms.append((torch.randn(B, 1, H, W, device="cuda") - 2).requires_grad_(True))
    gys.append(torch.randn(B, C, H, W, device="cuda"))


leaves = xs + ms + [p for m in mods for p in m.parameters()]


def step():
    for t in leaves:          # zero_grad(set_to_none=True): without it autograd ADDS into last step's gradients (24 extra kernels)
        t.grad = None
    ys = [m([x, k]) for m, x, k in zip(mods, xs, ms)]
    torch.autograd.backward(ys, gys)


for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 100
for _ in range(n): step()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / n
print(f"eager nn.Module step (3 levels, fwd+bwd through autograd): {el * 1e3:.3f} ms  = {B / el:.0f} img/s")

torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): step()
host = (time.perf_counter() - t0) / n          # enqueue time only (no sync yet)
torch.cuda.synchronize()
print(f"host enqueue time per step: {host * 1e3:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
