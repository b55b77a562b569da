"""Per-image step time vs batch (working set vs the 256 MB Infinity Cache): cfg2 widths at B = 8, 16, 32, 64."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
def timeit(g, n=200):
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B in (8, 16, 32, 64):
    bench.WORKLOADS["tmp"] = ("tmp", B, [(64, 80, 80), (128, 40, 40), (256, 20, 20)])
    plan, _, _ = bench.make_plan("tmp", dev, 1)
    g = plan.capture(lambda: (plan.forward(), plan.backward()))
    t = timeit(g)
    ws = plan.elements() * 4 * 4 / 1e6
    kt = bench.time_kernels(plan, 10)
    ks = {k: round(v, 1) for k, v in kt.items() if not k.startswith("_")}
    print(f"B={B:3d} working set {ws:6.0f} MB  step {t:7.1f} us  per image {t / B:6.2f} us  kernels {ks}", flush=True)
    del plan, g
# two half-batches back to back in one graph (sample-blocking) vs one full batch
bench.WORKLOADS["tmp"] = ("tmp", 16, [(64, 80, 80), (128, 40, 40), (256, 20, 20)])
pa, _, _ = bench.make_plan("tmp", dev, 1)
pb, _, _ = bench.make_plan("tmp", dev, 2)
g2 = pa.capture(lambda: (pa.forward(), pa.backward(), pb.forward(), pb.backward()))
print("2 x B=16 chunks in one graph: %.1f us per 32 images" % timeit(g2))
bench.WORKLOADS["tmp"] = ("tmp", 8, [(64, 80, 80), (128, 40, 40), (256, 20, 20)])
ps = [bench.make_plan("tmp", dev, s)[0] for s in range(4)]
g4 = ps[0].capture(lambda: [(p.forward(), p.backward()) for p in ps])
print("4 x B=8 chunks in one graph: %.1f us per 32 images" % timeit(g4))
