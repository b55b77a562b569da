"""Each launch of the step timed ALONE (the same launch 20x back to back) vs in step order: separates a kernel's own speed from what it
inherits from its predecessor (dirty lines of the previous kernel's output still draining from the Infinity Cache, cold L2).
    python tools/stage_alone.py [workload] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mga_yolo_amd import _lib

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
plan, desc, batch = bench.make_plan(wl, torch.device("cuda", 0), seed=1, dtype_name=dt)
Bs, Fs = _lib.BWD_STAGES, _lib.FWD_STAGES
plan.forward(); plan.backward(); torch.cuda.synchronize()
gate = plan.gate_active()
fold = plan.fold_active()
stages = [("fwd.pool", plan.forward, Fs["pool"], 1)]
stages += [("fwd.gate", plan.forward, Fs["chan"] | Fs["apply"], 2)] if gate else [("fwd.chan", plan.forward, Fs["chan"], 1), ("fwd.apply", plan.forward, Fs["apply"], 2)]
stages += [("bwd.reduce1+convT(fold)", plan.backward, Bs["reduce1"] | Bs["convT"] | _lib.BWD_FOLD, 2)] if fold else \
          [("bwd.reduce1", plan.backward, Bs["reduce1"], 2), ("bwd.convT", plan.backward, Bs["convT"], 0)]
if fold:
    stages += [("bwd.r12 (merged launch)", plan.backward, Bs["reduce1"] | Bs["convT"] | Bs["reduce2"] | Bs["wsa"] | _lib.BWD_FUSE | _lib.BWD_FOLD, 3)]
stages += [("bwd.reduce2+wsa", plan.backward, Bs["reduce2"] | Bs["wsa"] | _lib.BWD_FUSE, 1),
           ("bwd.reduce2 only", plan.backward, Bs["reduce2"], 1),
           ("bwd.apply+params", plan.backward, Bs["params"] | Bs["apply"] | _lib.BWD_FUSE, 3),
           ("bwd.apply only", plan.backward, Bs["apply"], 3)]
E = plan.elements() * (4 if dt == "f32" else 2)
print(f"{wl} {dt}: E = {E / 1e6:.1f} MB per feature tensor; gate={gate} fold={fold}")
for name, fn, mask, mult in stages:
    for _ in range(3):
        fn(mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        fn(mask)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"  {name:26s} alone {us:8.1f} us" + (f"   {mult * E / us / 1e3:7.0f} GB/s on {mult} E" if mult else ""))
plan.check_handoff()
