"""Where the host time of the eager (unchanged-trainer) path goes: the three MaskCBAM modules through autograd with the engine
kept on the calling thread, so cProfile sees the backward too.   python tools/eager_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mga_yolo_amd import MaskCBAM

torch.autograd.set_multithreading_enabled(False)
LV = [(64, 80, 80), (128, 40, 40), (256, 20, 20)]
B = 32
mods, xs, ms, gys = [], [], [], []
for C, H, W in LV:
    torch.manual_seed(0)
    mods.append(MaskCBAM(C).cuda())
    xs.append(torch.randn(B, C, H, W, device="cuda", requires_grad=True))
    ms.append((torch.randn(B, 1, H, W, device="cuda") - 2).requires_grad_(True))
    gys.append(torch.randn(B, C, H, W, device="cuda"))


def fwd():
    return [m([x, k]) for m, x, k in zip(mods, xs, ms)]


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return host * 1e6, (time.perf_counter() - t0) / n * 1e6


def nograd():
    with torch.no_grad():
        fwd()


leaves = xs + ms + [p for m in mods for p in m.parameters()]


def step():
    for t in leaves:          # zero_grad(set_to_none=True): without it autograd ADDS into last step's gradients (24 extra kernels)
        t.grad = None
    torch.autograd.backward(fwd(), gys)


for name, fn in (("forward + backward", step), ("forward, no_grad", nograd), ("forward, grad mode", fwd), ("forward + backward", step)):
    h, t = timeit(fn)
    print(f"{name:22s} host enqueue {h:7.1f} us   wall {t:7.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
