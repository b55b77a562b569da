"""Per-workgroup start / end stamps of the merged backward launch k_bwd_r12 (-DMGACBAM_TRACE build), by phase: k_bwd_reduce1 tiles,
transposed-conv tiles, dWsa tiles, k_bwd_reduce2 sweeps; and how many workgroups of each phase are alive over time.
    MGACBAM_LIB=$PWD/build/variants/libmgacbam_trace.so python tools/trace_r12.py [workload]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
L = __import__("mga_yolo_amd")._lib
S = L.BWD_STAGES
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
plan, desc, batch = bench.make_plan(wl, torch.device("cuda", 0), seed=1, dtype_name="f32")
buf = torch.zeros(16384 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    plan.forward(); plan.backward()
torch.cuda.synchronize()
plan.forward()
torch.cuda.synchronize()
os.environ["MGACBAM_TRACE_PTR"] = str(buf.data_ptr()); L.reload_env()
plan.backward(S["reduce1"] | S["convT"] | S["reduce2"] | S["wsa"] | L.BWD_FUSE | L.BWD_FOLD)      # the merged launch alone
torch.cuda.synchronize()
os.environ["MGACBAM_TRACE_PTR"] = ""; L.reload_env()
t = buf.cpu().numpy().reshape(-1, 16)
t = t[(t[:, 0] > 0) & (t[:, 14] > 0)]
t0 = t[:, 0].min()
names = {1: "k_bwd_reduce1 tiles", 2: "transposed-conv tiles", 3: "dWsa tiles", 4: "k_bwd_reduce2 sweeps"}
pc = lambda v: " ".join("%7.2f" % np.percentile(v, q) for q in (0, 10, 50, 90, 100))
print(f"{wl}: {len(t)} workgroups; launch length {(t[:, 10].max() - t0) / 100.0:.2f} us   (min p10 p50 p90 max, us)")
for ph in (1, 2, 3, 4):
    tt = t[t[:, 14] == ph]
    if not len(tt):
        continue
    st, en = (tt[:, 0] - t0) / 100.0, (tt[:, 10] - t0) / 100.0
    print(f"{names[ph]}: {len(tt)} workgroups\n  start  {pc(st)}\n  end    {pc(en)}\n  life   {pc(en - st)}")
end = (t[:, 10].max() - t0) / 100.0
print("alive at t (us):  " + "  ".join(f"{names[p].split()[0][:14]:>14s}" for p in (1, 2, 3, 4)))
for x in np.arange(0, end + 4, 4.0):
    row = []
    for ph in (1, 2, 3, 4):
        tt = t[t[:, 14] == ph]
        row.append(int((((tt[:, 0] - t0) / 100.0 <= x) & ((tt[:, 10] - t0) / 100.0 > x)).sum()))
    print(f"  t={x:5.1f}          " + "  ".join(f"{v:14d}" for v in row))
