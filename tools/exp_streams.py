"""Experiment: the three pyramid levels as three concurrent kernel chains (graph branches) vs one grouped launch per stage."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from mga_yolo_amd.plan import PyramidPlan
from mga_yolo_amd import MaskCBAM

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
plan, desc, batch = bench.make_plan(wl, dev, 1)

def timeit(g, n=200):
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

g1 = plan.capture(lambda: (plan.forward(), plan.backward()))
print(wl, "grouped (5 launches): %.1f us/step" % timeit(g1))

# per-level plans on three streams inside one graph
descs, batch, lv = bench.WORKLOADS[wl]
plans = []
for (C, H, W) in lv:
    torch.manual_seed(0)
    m = MaskCBAM(C)
    p = PyramidPlan([(batch, C, H, W)], [m.block_params()], [m.block_config()], device=dev)
    plans.append(p)
for l, p in enumerate(plans):
    p.x[0].copy_(plan.x[l]); p.mask[0].copy_(plan.mask[l]); p.gy[0].copy_(plan.gy[l])
streams = [torch.cuda.Stream(dev) for _ in plans]
def branches():
    cur = torch.cuda.current_stream(dev)
    for s, p in zip(streams, plans):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            p.forward(); p.backward()
    for s in streams:
        cur.wait_stream(s)
g3 = plans[0].capture(branches)
print(wl, "three concurrent chains (15 launches): %.1f us/step" % timeit(g3))
torch.cuda.synchronize()
for l, p in enumerate(plans):
    p.check_handoff()
    assert torch.allclose(p.gx[0], plan.gx[l], rtol=1e-4, atol=1e-5), l
# two branches: P3 alone, P4+P5 grouped
pb = PyramidPlan([(batch, *lv[1]), (batch, *lv[2])][0:2] and [(batch, lv[1][0], lv[1][1], lv[1][2]), (batch, lv[2][0], lv[2][1], lv[2][2])],
                 [plans[1].params[0], plans[2].params[0]], [plans[1].cfgs[0], plans[2].cfgs[0]], device=dev)
for i, l in enumerate((1, 2)):
    pb.x[i].copy_(plan.x[l]); pb.mask[i].copy_(plan.mask[l]); pb.gy[i].copy_(plan.gy[l])
def two():
    cur = torch.cuda.current_stream(dev)
    for s, p in zip(streams[:2], (plans[0], pb)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            p.forward(); p.backward()
    for s in streams[:2]:
        cur.wait_stream(s)
g2 = plans[0].capture(two)
print(wl, "two concurrent chains P3 | P4+P5 (10 launches): %.1f us/step" % timeit(g2))
# forward only comparisons
gf1 = plan.capture(plan.forward)
def fbr():
    cur = torch.cuda.current_stream(dev)
    for s, p in zip(streams, plans):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            p.forward()
    for s in streams:
        cur.wait_stream(s)
gf3 = plans[0].capture(fbr)
print(wl, "forward only: grouped %.1f us, three chains %.1f us" % (timeit(gf1), timeit(gf3)))
