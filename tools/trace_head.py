"""Poor man's thread trace of the mask-head GEMM kernels (needs a -DMGACBAM_TRACE build, see tools/trace_gate.py): thread 0 of every
workgroup records the 100 MHz wall clock at start / K loop done / K split summed / stores issued / stores complete.
    MGACBAM_LIB=$PWD/build/variants/libmgacbam_trace.so python tools/trace_head.py [fwd|gx] [workload]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mga_yolo_amd import _lib
from mga_yolo_amd.slice import SlicePlan
from mga_yolo_amd import MGAMaskHead, MaskCBAM

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
desc, batch, lv = bench.WORKLOADS[wl]
shapes, hidden, cps, cfgs, hss = [], [], [], [], []
for (C, H, W) in lv:
    torch.manual_seed(0)
    m = MaskCBAM(C); hid = C // 4; h = MGAMaskHead(C, hid)
    shapes.append((batch, C, H, W)); hidden.append(hid); cps.append(m.block_params()); cfgs.append(m.block_config()); hss.append(h.state_dict())
plan = SlicePlan(shapes, hidden, cps, cfgs, hss)
g = torch.Generator().manual_seed(7)
for l, (B, C, H, W) in enumerate(shapes):
    plan.x[l].copy_(torch.randn(B, C, H, W, generator=g)); plan.gy[l].copy_(torch.randn(B, C, H, W, generator=g))
for _ in range(3):
    plan.step()
torch.cuda.synchronize()
buf = torch.zeros(6 * 8192 * 16, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
if which == "fwd":
    plan.cbam.backward()                                        # the predecessor of the head forward in step order
else:
    plan.forward(); plan.cbam.backward()
torch.cuda.synchronize() if which != "fwd" else None
os.environ["MGACBAM_TRACE_PTR"] = str(buf.data_ptr()); _lib.reload_env()
if which == "fwd":
    _lib.check(plan.lib.mgahead_forward(plan._hf, plan.n, st), "fwd")
else:
    _lib.check(plan.lib.mgahead_backward(plan._hb, plan.n, st), "bwd")
torch.cuda.synchronize()
os.environ["MGACBAM_TRACE_PTR"] = ""; _lib.reload_env()
t = buf.cpu().numpy().reshape(-1, 16)
NAMES = {"gemm": ((0, "start"), (2, "K loop done"), (3, "K split summed"), (5, "stores issued"), (10, "stores complete")),
         "out": ((0, "start"), (1, "first pass staged"), (2, "all passes done"), (10, "end")),
         "gw": ((0, "start"), (1, "constants staged"), (2, "wave 0 pixel loop done"), (10, "end")),
         "act": ((0, "start"), (1, "g_logits staged"), (3, "z arrived"), (4, "wave 0 channels done"), (2, "all waves done"), (10, "end"))}
for base, name, kind in ((0, "fwd gemm pass 0 (one-tile levels)", "gemm"), (8192, "fwd gemm pass 1", "gemm"), (16384, "gx", "gemm"),
                         (24576, "k_head_out", "out"), (32768, "k_head_bwd_act", "act"), (40960, "k_head_bwd_gw", "gw")):
    tt = t[base:base + 8192]
    tt = tt[tt[:, 0] > 0]
    if not len(tt):
        continue
    t0 = tt[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    print(f"== {name}: {len(tt)} workgroups")
    print("%-18s %8s %8s %8s %8s %8s" % ("phase", "min", "p10", "p50", "p90", "max"))
    for s, n in NAMES[kind]:
        v = us(tt[:, s]); v = v[tt[:, s] > 0]
        if len(v):
            print("%-18s %8.2f %8.2f %8.2f %8.2f %8.2f" % (n, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    sl = [q for q, _ in NAMES[kind]]
    for a, b, n in [(sl[i], sl[i + 1], f"{NAMES[kind][i][1]} -> {NAMES[kind][i + 1][1]}") for i in range(len(sl) - 1)] + [(0, 10, "workgroup life")]:
        ok = (tt[:, a] > 0) & (tt[:, b] > 0)
        d = (tt[ok, b] - tt[ok, a]) / 100.0
        if len(d):
            print("  %-42s median %6.2f  p90 %6.2f" % (n, np.median(d), np.percentile(d, 90)))
    hw = tt[:, 15]; cu = (hw >> 32) * 1000 + ((hw >> 8) & 0xf) + 16 * ((hw >> 13) & 0x7)       # XCC id, CU id, SE id
    print("  distinct (xcc, se, cu): %d; workgroups alive at the median start time: %d" % (len(np.unique(cu)),
          int(((tt[:, 0] <= np.median(tt[:, 0])) & (tt[:, 10] >= np.median(tt[:, 0]))).sum())))
