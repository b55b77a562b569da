"""Poor man's thread trace of k_gate (needs a -DMGACBAM_TRACE build: python -c "from mga_yolo_amd import build;
build.build(defines=['-DMGACBAM_TRACE'], out='build/variants/libmgacbam_trace.so')", then run with
MGACBAM_LIB=$PWD/build/variants/libmgacbam_trace.so).  Thread 0 of every workgroup records the 100 MHz wall clock at phase
boundaries; this prints, per phase, when workgroups reach it (percentiles over workgroups, us since the first start) and the
per-CU residency."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NAMES_FWD = ["start", "ca flag", "ca in LDS", "x scanned", "planes stored", "published", "neighbours", "staged", "conv", "stores issued",
             "stores done"]
NAMES_BWD = ["start", "g_h", "terms", "loop done", "-", "-", "-", "-", "-", "-", "stores done"]   # k_bwd_apply
NAMES_POOL = ["start", "-", "-", "sweep done", "-", "-", "-", "-", "reduced", "-", "stores done"]  # k_pool


def main():
    plan, desc, batch = bench.make_plan(sys.argv[2] if len(sys.argv) > 2 else "cfg2", torch.device("cuda", 0), seed=1, dtype_name="f32")
    nblk = 16384
    buf = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
    which = sys.argv[1] if len(sys.argv) > 1 else "fwd"       # fwd: k_gate, bwd: k_bwd_apply
    NAMES = {"fwd": NAMES_FWD, "bwd": NAMES_BWD, "pool": NAMES_POOL}[which]
    L = __import__("mga_yolo_amd")._lib
    S = L.BWD_STAGES
    for _ in range(3):
        plan.forward(); plan.backward()
    torch.cuda.synchronize()
    if which == "fwd":
        plan.forward(1)
        torch.cuda.synchronize()
        os.environ["MGACBAM_TRACE_PTR"] = str(buf.data_ptr()); L.reload_env()
        plan.forward(6)
    elif which == "pool":
        plan.backward()
        torch.cuda.synchronize()
        os.environ["MGACBAM_TRACE_PTR"] = str(buf.data_ptr()); L.reload_env()
        plan.forward(1)
    else:
        plan.forward()
        plan.backward(S["reduce1"]); plan.backward(S["convT"]); plan.backward(S["reduce2"] | S["wsa"] | 64)
        torch.cuda.synchronize()
        os.environ["MGACBAM_TRACE_PTR"] = str(buf.data_ptr()); L.reload_env()
        plan.backward(S["params"] | S["apply"] | 64)
    torch.cuda.synchronize()
    os.environ["MGACBAM_TRACE_PTR"] = ""; L.reload_env()
    t = buf.cpu().numpy().reshape(nblk, 16)
    used = t[:, 0] > 0
    ids = np.nonzero(used)[0]
    t = t[used]
    t0 = t[:, 0].min()
    role = t[:, 10] == 0
    nblk_seen = used.sum()
    print(f"{used.sum()} workgroups traced, {role.sum()} role workgroups; clock 100 MHz -> us")
    us = lambda v: (v - t0) / 100.0
    tiles = t[~role]
    print("%-16s %8s %8s %8s %8s %8s" % ("phase", "min", "p10", "p50", "p90", "max"))
    for s, n in enumerate(NAMES):
        if n == "-":
            continue
        v = us(tiles[:, s])
        print("%-16s %8.2f %8.2f %8.2f %8.2f %8.2f" % (n, v.min(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
    print("per-workgroup phase durations (us): median / p90")
    live = [s for s, n in enumerate(NAMES) if n != "-"]
    for s0, s1 in zip(live[:-1], live[1:]):
        d = (tiles[:, s1] - tiles[:, s0]) / 100.0
        print("  %-14s -> %-14s %7.2f %7.2f" % (NAMES[s0], NAMES[s1], np.median(d), np.percentile(d, 90)))
    r = t[role]
    if len(r):
        last = 5 if which == "fwd" else 9
        print("role workgroups: start p50 %.2f, done p50 %.2f p90 %.2f max %.2f" % (
            np.median(us(r[:, 0])), np.median(us(r[:, last])), np.percentile(us(r[:, last]), 90), us(r[:, last]).max()))
    hw = tiles[:, 15]
    xcc = hw >> 32
    cu = (hw & 0xFFFFFFFF)
    cu_id = (cu >> 8) & 0xF
    se_id = (cu >> 13) & 0x7
    key = xcc * 1000 + se_id * 16 + cu_id
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"{len(uniq)} distinct (xcc,se,cu); workgroups per CU: min {cnt.min()} median {np.median(cnt)} max {cnt.max()}")
    # hardware slot ids of the workgroups that start in the first 2 us (the first resident round): is there one of each per CU?
    first = us(tiles[:, 0]) < 2.0
    wave_id, simd_id, tg_id = cu & 0xF, (cu >> 4) & 0x3, (cu >> 16) & 0xF
    for name, v in (("wave_id", wave_id), ("simd_id", simd_id), ("tg_id", tg_id)):
        vals, c = np.unique(v[first], return_counts=True)
        print(f"first-round {name}: " + " ".join(f"{a}:{b}" for a, b in zip(vals, c)))
    per_cu = {}
    for k_, w_, t_ in zip(key[first], wave_id[first], tg_id[first]):
        per_cu.setdefault(k_, []).append((int(w_), int(t_)))
    combos = {}
    for v in per_cu.values():
        combos[tuple(sorted(v))] = combos.get(tuple(sorted(v)), 0) + 1
    print("first-round (wave_id, tg_id) sets per CU:", sorted(combos.items(), key=lambda kv: -kv[1])[:8])
    # concurrency over time: how many tile workgroups are in [start, stores done) at each us
    life = np.stack([us(tiles[:, 0]), us(tiles[:, 10])], 1)
    for tt in range(0, int(life[:, 1].max()) + 1, 4):
        n = ((life[:, 0] <= tt) & (life[:, 1] > tt)).sum()
        a_, b_ = {"fwd": (3, 8), "bwd": (2, 3), "pool": (3, 8)}[which]          # bwd: "loading" = prologue, "chain" = streaming loop
        nload = ((us(tiles[:, 0]) <= tt) & (us(tiles[:, a_]) > tt)).sum()
        nchain = ((us(tiles[:, a_]) <= tt) & (us(tiles[:, b_]) > tt)).sum()
        nstore = ((us(tiles[:, b_]) <= tt) & (us(tiles[:, 10]) > tt)).sum()
        print(f"  t={tt:3d} us: resident {n:5d}  loading {nload:5d}  chain {nchain:5d}  storing {nstore:5d}")
    np.save("gpurun_out/trace_gate.npy", np.concatenate([ids[:, None], t], 1))


if __name__ == "__main__":
    main()
