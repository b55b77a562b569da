"""What the role workgroups cost inside their host launches: k_bwd_reduce2 with / without the dWsa-partial roles, k_bwd_apply with /
without the parameter-gradient roles, and the role kernels alone (event-timed single launches after the preceding stages).
Round 1, config 2: reduce2 28.1 / 22.3 us (roles alone 13.5), apply 56.0 / 53.4 us (roles alone 16.7)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from mga_yolo_amd import _lib
plan, desc, batch = bench.make_plan("cfg2", torch.device("cuda", 0), seed=1, dtype_name="f32")
S = _lib.BWD_STAGES
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        plan.backward(S["reduce1"]); plan.backward(S["convT"])
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return sum(ts[:n // 2]) / (n // 2)
plan.forward(); plan.backward()
print("reduce2 + wsa roles :", round(timeit(lambda: plan.backward(S["reduce2"] | S["wsa"] | _lib.BWD_FUSE)), 2))
print("reduce2 alone       :", round(timeit(lambda: plan.backward(S["reduce2"])), 2))
print("wsa alone           :", round(timeit(lambda: plan.backward(S["wsa"])), 2))
print("apply + params roles:", round(timeit(lambda: plan.backward(S["params"] | S["apply"] | _lib.BWD_FUSE)), 2))
print("apply alone         :", round(timeit(lambda: plan.backward(S["apply"])), 2))
print("params alone        :", round(timeit(lambda: plan.backward(S["params"])), 2))
