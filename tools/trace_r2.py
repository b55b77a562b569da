"""Per-workgroup start / end stamps of k_bwd_reduce2 (-DMGACBAM_TRACE build): sweep workgroups vs the dWsa role workgroups.
    MGACBAM_LIB=$PWD/build/variants/libmgacbam_trace.so python tools/trace_r2.py [workload]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
L = __import__("mga_yolo_amd")._lib
S = L.BWD_STAGES
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
plan, desc, batch = bench.make_plan(wl, torch.device("cuda", 0), seed=1, dtype_name="f32")
buf = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    plan.forward(); plan.backward()
torch.cuda.synchronize()
plan.forward()
plan.backward(S["reduce1"]); plan.backward(S["convT"])
torch.cuda.synchronize()
os.environ["MGACBAM_TRACE_PTR"] = str(buf.data_ptr()); L.reload_env()
plan.backward(S["reduce2"] | S["wsa"] | 64)
torch.cuda.synchronize()
os.environ["MGACBAM_TRACE_PTR"] = ""; L.reload_env()
t = buf.cpu().numpy().reshape(-1, 16)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
role = t[:, 10] == 0
for name, tt, end in (("sweep", t[~role], 10), ("dWsa roles", t[role], 5)):
    if not len(tt):
        continue
    st, en = (tt[:, 0] - t0) / 100.0, (tt[:, end] - t0) / 100.0
    pc = lambda v: " ".join("%7.2f" % np.percentile(v, q) for q in (0, 10, 50, 90, 100))
    print(f"{name}: {len(tt)} workgroups\n  start  {pc(st)}\n  end    {pc(en)}\n  life   {pc(en - st)}   (min p10 p50 p90 max, us)")
