"""Soak test of the in-launch hand-off: N graph replays of the full step at config 2 (default 20 000 = 40 000 k_gate launches'
worth of flag generations incl. the warm-ups), then: time-out words still 0, every tile flag == number of fused calls, outputs
bit-identical to the first replay.    python tools/soak_gate.py [n_replays]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
plan, desc, batch = bench.make_plan("cfg2", torch.device("cuda", 0), seed=1, dtype_name="f32")
g = plan.capture(lambda: (plan.forward(), plan.backward()))
g.replay(); torch.cuda.synchronize()
ref = [t.clone() for t in plan.y] + [t.clone() for t in plan.gx] + [plan.grad_bucket.clone()]
t0 = time.perf_counter()
for i in range(n):
    g.replay()
    if i % 2000 == 1999:
        torch.cuda.synchronize()
        print(f"  {i + 1} replays, {(time.perf_counter() - t0) / (i + 1) * 1e6:.1f} us/step", flush=True)
torch.cuda.synchronize()
ok = True
for a, b in zip(ref, list(plan.y) + list(plan.gx) + [plan.grad_bucket]):
    ok &= bool(torch.equal(a, b))
for l, (B, C, H, W) in enumerate(plan.shapes):
    s = plan.ctx_view(l)["sync"]
    nf = B * ((H * W + 15) // 16 + 1)
    flags = s[:nf]
    calls = int(flags.max())
    ok &= int(s[nf:nf + 4].abs().sum()) == 0 and set(flags.unique().tolist()) <= {0, calls}
    bw = s[nf + 4 + B:]                                             # MGACBAM_BWD_FOLD's tile / conv-tile counters: one bump per folded launch
    ok &= set(bw.unique().tolist()) <= {0, int(bw.max())}
    print(f"level {l}: fused calls {calls}, status words {s[nf:nf + 4].tolist()}")
print("soak:", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
