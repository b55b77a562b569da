#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2]

A *step* is one pass of the hot path over one batch: MaskCBAM forward + backward at P3, P4 and P5 (YOLOv8n channel
widths, 640x640 input -> 80/40/20 px feature maps, batch 32 per GPU, fp32, random masks = BASELINE.json configs[1]),
followed -- when N > 1 -- by the data-parallel exchange of the block's parameter gradients (one flat bucket, RCCL
all-reduce on a side stream; it overlaps the NEXT step's parameter-free prefix (k_pool) and is joined before the first
kernel that reads a parameter, which is the DDP contract: averaged gradients before the next use of the weights).  Inputs are synthetic and already resident in HBM when timing
starts.  For N > 1 the driver starts one process per GPU with torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* are read from the environment.  Rank 0 prints ONE JSON line:

  value        images/s over all N GPUs = N * batch / step time (max over ranks)             "scaling": "weak"
  roofline     the dominant kernel: algorithmic bytes per launch / its mean duration (events on the launch stream)
  kernels      the same for every HBM-bound kernel, plus the whole step against SURVEY 8d's 8*E*4 bytes
  cpu_baseline the oracle's eager-op form (same op sequence as the reference's PyTorch-CPU path) timed on this
               box's host cores on a bounded sample (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s peak, ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (description, per-GPU batch, [(C, H, W) for P3, P4, P5])
    "cfg2": ("YOLOv8n+MGA-CBAM P3/P4/P5, 32x640x640 synthetic per GPU, fp32 (BASELINE.json configs[1])", 32,
             [(64, 80, 80), (128, 40, 40), (256, 20, 20)]),
    # single-level diagnostics (not benchmark lines)
    "p3": ("diagnostic: P3 only of cfg2", 32, [(64, 80, 80)]),
    "p4": ("diagnostic: P4 only of cfg2", 32, [(128, 40, 40)]),
    "p5": ("diagnostic: P5 only of cfg2", 32, [(256, 20, 20)]),
    "cfg1": ("YOLOv8n+MGA-CBAM P3/P4/P5, 2x640x640 (BASELINE.json configs[0] shapes)", 2,
             [(64, 80, 80), (128, 40, 40), (256, 20, 20)]),
    "cfg3": ("YOLOv8s+MGA-CBAM P3/P4/P5, 32x640x640 per GPU (BASELINE.json configs[2] shapes)", 32,
             [(128, 80, 80), (256, 40, 40), (512, 20, 20)]),
    "cfg4": ("YOLOv8m+MGA-CBAM P3/P4/P5, 8x1280x1280 per GPU (BASELINE.json configs[3] shapes, > Infinity Cache)", 8,
             [(256, 160, 160), (512, 80, 80), (512, 40, 40)]),
}

# whole-model fp32 gradient payload per step of the model each workload stands for (parameters x 4 B), and its scale letter
MODEL_GRAD_MB = {"cfg2": 11.9, "cfg1": 11.9, "p3": 11.9, "p4": 11.9, "p5": 11.9, "cfg3": 43.7, "cfg4": 93.7}
MODEL_SCALE = {"cfg3": "s", "cfg4": "m"}

# algorithmic bytes of each HBM-bound kernel in units of E*w (feature-sized tensors it must read or write once)
FWD_KERNEL_E = {"pool": 1, "chan": 1, "apply": 2, "gate": 2}   # k_gate = chan + apply with x resident: read x once, write y
BWD_KERNEL_E = {"reduce1": 2, "reduce2": 1, "apply": 3, "r12": 3}   # r12 = reduce1 + reduce2 in one launch (k_bwd_r12): x, gy, x
KERNEL_SYMBOL_FOLD = {"bwd.reduce1": "k_bwd_reduce1_fold"}
KERNEL_SYMBOL = {"bwd.r12": "k_bwd_r12", "fwd.pool": "k_pool", "fwd.chan": "k_chan", "fwd.apply": "k_apply", "fwd.gate": "k_gate", "bwd.reduce1": "k_bwd_reduce1",
                 "bwd.reduce2": "k_bwd_reduce2", "bwd.apply": "k_bwd_apply", "bwd.convT": "k_bwd_convT"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="issue every launch from Python instead of replaying hipGraphs")
    ap.add_argument("--steps-per-graph", type=int, default=1,
                    help="N = 1 only: steps recorded per hipGraph (diagnostic of the launch gap; the headline uses 1 = one replay per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager", action="store_true", help="skip the eager-module (unchanged-trainer) timing")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the bounded CPU-baseline sample (>= 50 iterations at config 2 need ~15 s)")
    ap.add_argument("--kernel-reps", type=int, default=30)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16", "bf16"], help="element type of x / y / gy / gx (headline = f32)")
    ap.add_argument("--grad-payload-mb", type=float, default=None,
                    help="N > 1: fp32 gradient bytes of the layers OUTSIDE the hot path that ride in the same all-reduce every step, in 25 MB "
                         "buckets as DDP cuts them (default: the workload's whole-model figure, SURVEY 2: n 11.9, s 43.7, m 93.7; 0 = the blocks' own bucket only)")
    ap.add_argument("--no-harness", action="store_true", help="skip the train-step harness (tools/harness.py, extra key train_harness)")
    ap.add_argument("--harness-steps", type=int, default=8)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only for rehearsing the multi-rank code path on one GPU)")
    return ap.parse_args()


def relaunch_distributed(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU with torch.distributed.run (child process)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def make_plan(workload, device, seed, dtype_name="f32"):
    import torch
    from mga_yolo_amd import MaskCBAM
    from mga_yolo_amd.plan import PyramidPlan
    desc, batch, lv = WORKLOADS[workload]
    shapes, params, cfgs = [], [], []
    for (C, H, W) in lv:
        torch.manual_seed(0)                                   # default init, seed 0
        m = MaskCBAM(C)
        shapes.append((batch, C, H, W)); params.append(m.block_params()); cfgs.append(m.block_config())
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype_name]
    plan = PyramidPlan(shapes, params, cfgs, dtype=dt, device=device, with_mask=True, want_gmask=True)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for l, (B, C, H, W) in enumerate(shapes):                  # SiLU-shaped features, sparse (vessel-like) masks, N(0,1) upstream grads
        plan.x[l].copy_(torch.nn.functional.silu(torch.randn(B, C, H, W, generator=g)))
        plan.mask[l].copy_(torch.randn(B, 1, H, W, generator=g) - 2.0)
        plan.gy[l].copy_(torch.randn(B, C, H, W, generator=g))
    return plan, desc, batch


def time_kernels(plan, reps):
    """Mean duration of each of the step's launches, measured IN STEP ORDER (so every kernel sees the cache state the
    previous one leaves, as in the real step): the step is issued eagerly `reps` times with an event before and after each
    library call on the launch stream; elapsed(before, after) brackets exactly one kernel launch."""
    import torch
    from mga_yolo_amd import _lib
    Bs, Fs = _lib.BWD_STAGES, _lib.FWD_STAGES
    fwd = [("fwd.pool", plan.forward, Fs["pool"])]
    plan.forward()
    if plan.gate_active():
        fwd += [("fwd.gate", plan.forward, Fs["chan"] | Fs["apply"])]     # one x-resident launch (plan adds FWD_FUSE)
    else:
        fwd += [("fwd.chan", plan.forward, Fs["chan"]), ("fwd.apply", plan.forward, Fs["apply"])]
    folded = plan.fold_active()
    fold_bit = _lib.BWD_FOLD if plan.fold_backward else 0
    wsa_tail = os.environ.get("MGACBAM_WSA_TAIL", "0") == "1"
    # the real step's backward (one mgacbam_backward call) is TWO launches when the fold applies: k_bwd_r12 (k_bwd_reduce1 tiles,
    # transposed-conv tiles, dWsa tiles and k_bwd_reduce2 sweeps with in-launch hand-offs) and k_bwd_apply
    merged = folded and all(c.k == 7 for c in plan.cfgs) and os.environ.get("MGACBAM_BWD_MERGE", "1") != "0" and not wsa_tail
    if merged:
        seq = fwd + [("bwd.r12", plan.backward, Bs["reduce1"] | Bs["convT"] | Bs["reduce2"] | Bs["wsa"] | _lib.BWD_FUSE | _lib.BWD_FOLD),
                     ("bwd.apply", plan.backward, Bs["params"] | Bs["apply"] | _lib.BWD_FUSE | fold_bit)]
    elif folded:                 # the transposed conv rides at the end of the k_bwd_reduce1 launch (one launch, as in the real step)
        bwd = [("bwd.reduce1", plan.backward, Bs["reduce1"] | Bs["convT"] | _lib.BWD_FOLD)]
    else:
        bwd = [("bwd.reduce1", plan.backward, Bs["reduce1"]), ("bwd.convT", plan.backward, Bs["convT"])]
    # the two fused launches: streaming workgroups + role workgroups.  With the zero-filled hand-off state (fold_bit) the dWsa tile partials
    # are the LAST workgroups of the k_bwd_apply launch, as in the real step (one mgacbam_backward call); else leading roles of k_bwd_reduce2
    wsa_in_apply = Bs["wsa"] if (fold_bit and wsa_tail) else 0
    if not merged:
        seq = fwd + bwd + [
               ("bwd.reduce2", plan.backward, Bs["reduce2"] | (Bs["wsa"] ^ wsa_in_apply) | _lib.BWD_FUSE),
               ("bwd.apply", plan.backward, Bs["params"] | Bs["apply"] | wsa_in_apply | _lib.BWD_FUSE | fold_bit)]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)] for _ in range(reps)]
    for _ in range(3):
        for _, fn, mask in seq:
            fn(mask)
    torch.cuda.synchronize()
    for r in range(reps):
        ev[r][0].record()
        for k, (_, fn, mask) in enumerate(seq):
            fn(mask)
            ev[r][k + 1].record()
    torch.cuda.synchronize()
    # the same eager step without the inner events: the difference, spread over the launches, is what the event records
    # themselves cost on the stream (it is not kernel time and is subtracted)
    o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    o0.record()
    for r in range(reps):
        for _, fn, mask in seq:
            fn(mask)
    o1.record()
    torch.cuda.synchronize()
    t_outer = o0.elapsed_time(o1) * 1e3 / reps
    out, fast = {}, {}
    keep = max(1, (3 * reps) // 4)
    for k, (name, _, _) in enumerate(seq):
        ts = sorted(ev[r][k].elapsed_time(ev[r][k + 1]) * 1e3 for r in range(reps))
        out[name] = sum(ts) / len(ts)                           # plain mean over all repetitions (what roofline.achieved uses)
        fast[name] = sum(ts[:keep]) / keep                      # mean of the fastest 75 % (extra key: drops host hiccups)
    pad = max(0.0, (sum(out.values()) - t_outer) / len(seq))     # what the inner event records cost on the stream (reported, NOT subtracted)
    out["_event_pad_us"] = pad
    out["_folded"] = folded
    out["_fast75"] = fast
    return out


def eager_module_step(workload, device, dtype_name, steps=60, warmup=10, engine_inline=False):
    """The path the reference's UNCHANGED layer loop takes (mga_yolo/model/model.py:57-64): three `MaskCBAM` modules called one
    after the other through autograd (one library call forward and one backward PER LEVEL, every launch issued from Python),
    no PyramidPlan, no hipGraph.  Returns ms per forward+backward step over the three levels."""
    import torch
    from mga_yolo_amd import MaskCBAM
    desc, batch, lv = WORKLOADS[workload]
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype_name]
    g = torch.Generator().manual_seed(99)
    mods, data = [], []
    for (C, H, W) in lv:
        torch.manual_seed(0)
        mods.append(MaskCBAM(C).to(device))
        x = torch.nn.functional.silu(torch.randn(batch, C, H, W, generator=g)).to(device, dt).requires_grad_(True)
        m = (torch.randn(batch, 1, H, W, generator=g) - 2.0).to(device).requires_grad_(True)
        gy = torch.randn(batch, C, H, W, generator=g).to(device, dt)
        data.append((x, m, gy))

    leaves = [t for x, m, _ in data for t in (x, m)] + [p for mod in mods for p in mod.parameters()]

    def step():
        for t in leaves:                                         # optimizer.zero_grad(set_to_none=True), as the reference trainer does every
            t.grad = None                                        # step (otherwise autograd adds into the old gradients: 24 extra kernels)
        ys = [mod([x, m]) for mod, (x, m, _) in zip(mods, data)]
        torch.autograd.backward(ys, [gy for _, _, gy in data])
    # engine_inline: autograd's engine kept on the calling thread.  By default backward() hands the graph to a device worker thread and
    # waits; that wake-up (~0.2 ms here) is paid once per backward() call -- for the WHOLE model in a real training step -- so the
    # inline figure is the blocks' marginal cost inside a trainer, the default figure what a stand-alone three-node graph costs
    prev = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(not engine_inline)
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / steps
    finally:
        torch.autograd.set_multithreading_enabled(prev)


def slice_step(workload, device, steps=100, warmup=10, dtype_name="f32"):
    """SURVEY 8d images/s definition (2), in scope: the LAYER-LOOP SLICE of the training step (mga_yolo_amd/slice.py) -- three mask heads,
    three MaskCBAM blocks, the multi-scale segmentation loss, the Kendall combine and the backward of all of it, replayed from one
    hipGraph.  Backbone / neck / Detect / detection loss are out of scope: they enter as given tensors (dL/d refined, det_loss)."""
    import torch
    from mga_yolo_amd import MGAMaskHead, MaskCBAM
    from mga_yolo_amd.slice import SlicePlan
    desc, batch, lv = WORKLOADS[workload]
    shapes, hidden, cps, cfgs, hss = [], [], [], [], []
    for (C, H, W) in lv:
        torch.manual_seed(0)
        m = MaskCBAM(C)
        hid = max(8, ((C // 4) + 7) // 8 * 8)                    # yolov8_cbam.yaml: MGAMaskHead [C*4, C] width-scaled -> hidden = C / 4
        h = MGAMaskHead(C, hid)
        shapes.append((batch, C, H, W)); hidden.append(hid); cps.append(m.block_params()); cfgs.append(m.block_config()); hss.append(h.state_dict())
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype_name]
    plan = SlicePlan(shapes, hidden, cps, cfgs, hss, device=device, dtype=dt)
    g = torch.Generator(device="cpu").manual_seed(7)
    for l, (B, C, H, W) in enumerate(shapes):
        plan.x[l].copy_(torch.nn.functional.silu(torch.randn(B, C, H, W, generator=g)))
        plan.gy[l].copy_(torch.randn(B, C, H, W, generator=g))
        plan.targets[l].copy_((torch.rand(B, 1, H, W, generator=g) > 0.9).float())
    plan.det_loss.copy_(torch.tensor([1.0, 0.5, 1.5]))
    graph = plan.capture(plan.step)
    for _ in range(warmup):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    plan.check_handoff()
    E = sum(B * C * H * W for B, C, H, W in shapes) * 4
    # algorithmic bytes of the slice: MaskCBAM 8 E; head forward reads x (1 E) + writes / reads z (2 * E/4); head backward reads x (1 E),
    # accumulates gx (read + write 2 E), g_a / z traffic 4 * E/4
    return dict(ms_per_step=round(ms, 4), images_per_s=round(batch / (ms * 1e-3), 1), hidden=hidden, launches=plan.launches(),
                note="mask heads + MaskCBAM + seg loss + Kendall, fwd + bwd, one hipGraph; dL/d(refined) and det_loss are given")


def host_cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    quota = None
    try:                                                          # cgroup v2 CPU quota of this container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    usable = min(x for x in (logical, affinity, quota) if x)
    return dict(model=model, logical_cpus=logical, affinity=affinity, cgroup_quota=quota, usable=usable)


def cpu_baseline(workload, seconds):
    """The oracle's eager-op form = the reference's PyTorch-CPU op sequence, fwd+bwd over the three levels, timed on this box's
    host cores (BASELINE.md section 3 protocol: all usable cores, warm-up, >= 50 timed iterations unless the time budget runs
    out, median + p10 / p90)."""
    import torch
    from oracle import maskcbam_oracle as O
    desc, batch, lv = WORKLOADS[workload]
    info = host_cpu_info()
    cores = info["usable"]
    torch.set_num_threads(cores)
    sample_batch = batch
    g = torch.Generator().manual_seed(1234)
    data = []
    for (C, H, W) in lv:
        x = torch.nn.functional.silu(torch.randn(sample_batch, C, H, W, generator=g))
        mask = torch.randn(sample_batch, 1, H, W, generator=g) - 2.0
        gy = torch.randn(sample_batch, C, H, W, generator=g)
        data.append((x, mask, gy, O.Params.default_init(C)))
    cfg = O.Config()

    def one():
        for x, mask, gy, p in data:
            O.reference_form_step(x, mask, p, cfg, gy)
    for _ in range(3):                                           # warm-up
        one()
    ts = []
    t_start = time.perf_counter()
    while len(ts) < 50 and (time.perf_counter() - t_start) < seconds:
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
    el = sum(ts)
    srt = sorted(ts)
    pct = lambda q: srt[min(len(srt) - 1, int(q * len(srt)))]
    med = pct(0.5)
    return dict(value=round(sample_batch / med, 2), unit="images/s", cores=cores, kind="port",
                cpu_model=info["model"], logical_cpus=info["logical_cpus"], affinity=info["affinity"], cgroup_quota=info["cgroup_quota"],
                iterations=len(ts), ms_median=round(med * 1e3, 2), ms_p10=round(pct(0.1) * 1e3, 2), ms_p90=round(pct(0.9) * 1e3, 2),
                images_per_s_mean=round(sample_batch * len(ts) / el, 2),
                sample=f"{len(ts)} fwd+bwd steps (median) of the oracle's eager-op form (oracle/maskcbam_oracle.py:reference_form_step) "
                       f"over P3+P4+P5 at batch {sample_batch}, torch CPU fp32, {cores} threads on {info['model']}, {el:.1f} s")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(relaunch_distributed(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    from mga_yolo_amd import _lib
    from mga_yolo_amd.dp import GradExchange

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    if world > ndev and not os.environ.get("MGACBAM_BENCH_KEEP_HANDOFF"):
        # rehearsal only (several ranks on ONE card): the CUs are shared, so the co-residency the in-launch hand-offs are sized for
        # does not hold -- run the three-launch forward / unfolded backward (what `MGACBAM_FUSE_FWD=0` selects)
        os.environ["MGACBAM_FUSE_FWD"] = "0"
        os.environ["MGACBAM_FOLD_BWD"] = "0"
    plan, desc, batch = make_plan(args.workload, device, seed=1234 + rank, dtype_name=args.dtype)
    # N > 1: what DDP puts through the collective every step is the WHOLE model's gradient (U/engine/trainer.py:366-367), of which the
    # blocks' own bucket is 47 KB: the rest rides along as payload buckets, so that the exchange the step has to hide is the real one
    from mga_yolo_amd.dp import payload_buckets
    payload_mb = MODEL_GRAD_MB.get(args.workload, 0.0) if args.grad_payload_mb is None else args.grad_payload_mb
    payload = payload_buckets(int(payload_mb * 1e6) if world > 1 else 0, device)
    exchange = GradExchange([plan.grad_bucket] + payload)        # no-op when world == 1

    S = _lib.FWD_STAGES

    def part_a():                # parameter-free prefix of the step: masked pooling
        plan.forward(S["pool"])

    def part_b():                # everything that reads parameters: rest of forward, whole backward (5 launches/step in all)
        plan.forward(S["chan"] | S["apply"])
        plan.backward()

    def whole():                 # N=1: one library call each way (the forward is ONE launch when the plan fuses it)
        plan.forward()
        plan.backward()

    if args.no_graph:
        run_a, run_b, run_all = part_a, part_b, whole
    elif world == 1:
        g_all = plan.capture(whole)          # one graph per step: a graph boundary costs ~8 us on the device (rocprof trace)
        run_a = run_b = None
        run_all = g_all.replay
        U = max(1, args.steps_per_graph)
        g_multi = plan.capture(lambda: [whole() for _ in range(U)]) if U > 1 else None
    else:
        ga, gb = plan.capture(part_a), plan.capture(part_b)   # split only where the gradient exchange has to be joined
        run_a, run_b, run_all = ga.replay, gb.replay, None

    def step():
        if world == 1:
            run_all()
            return
        run_a()                  # overlaps the all-reduce started at the end of the previous step
        exchange.finish()        # averaged gradients of the previous step are complete before any parameter is read
        run_b()
        exchange.start()         # all-reduce of this step's flat gradient bucket on a side stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    for _ in range(args.warmup):
        step()
    fence()
    gc.collect()
    gc.disable()                 # a collector pause inside a 4-40 ms timed region is host noise, not the path being measured
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                 # extra key only (ms_per_step_device): the same K steps by the device's clock, without the host's wake-up at both ends
    if world == 1 and not args.no_graph and args.steps_per_graph > 1:     # diagnostic: exactly K steps, U per replay + the remainder
        for _ in range(args.steps // U):
            g_multi.replay()
        for _ in range(args.steps % U):
            step()
    else:
        for _ in range(args.steps):
            step()
    exchange.finish()            # the last step's exchange is inside the timed region
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    ms_device = ev0.elapsed_time(ev1) / args.steps
    gc.enable()
    plan.check_handoff()         # raises if any in-launch hand-off of the run timed out (the step would have been wrong AND slow)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * batch * args.steps / elapsed

    # ---- per-kernel durations on this GPU, isolated launches on the same stream -------------------------------
    kt = time_kernels(plan, args.kernel_reps)
    E = plan.elements()
    w = 4 if args.dtype == "f32" else 2
    kernels = {}
    event_pad = kt.pop("_event_pad_us")
    fast75 = kt.pop("_fast75")
    symbols = dict(KERNEL_SYMBOL, **(KERNEL_SYMBOL_FOLD if kt.pop("_folded") else {}))   # (bwd.r12 has its own entry)
    for name, us in kt.items():
        side, k = name.split(".")
        mult = (FWD_KERNEL_E if side == "fwd" else BWD_KERNEL_E).get(k)
        ent = dict(us=round(us, 2), us_fastest75=round(fast75[name], 2), launches=1, symbol=symbols[name])
        if mult:
            ent["alg_bytes"] = mult * E * w
            ent["GBps"] = round(mult * E * w / us / 1e3, 1)
        kernels[name] = ent
    dom = max((n for n in kernels if "alg_bytes" in kernels[n]), key=lambda n: kernels[n]["us"])
    # roofline.traffic = HBM bytes per launch from the PMC counters: they need rocprofv3 passes of their own (FETCH_SIZE / WRITE_SIZE,
    # MI355X_MICROARCH.md), so the figure is NOT measured in this run -- it is read from the committed summary of the last
    # profiling run and tagged with where it came from
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(args.workload if args.dtype == "f32" else f"{args.workload}_{args.dtype}", {}).get(symbols[dom])
            src = tj.get("_source")
            traffic_source = src.get(args.workload) if isinstance(src, dict) else src
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel=symbols[dom], stage=dom, achieved=kernels[dom]["GBps"], peak=HBM_PEAK_GBPS, unit="GB/s",
                    frac=round(kernels[dom]["GBps"] / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_source,
                    alg_bytes_per_launch=kernels[dom]["alg_bytes"], us=kernels[dom]["us"],
                    note="one launch covers P3+P4+P5; duration = plain mean over the repetitions of the elapsed time between an event recorded before and one after the launch, on the launch stream, with the step issued in order")
    step_alg = 8 * E * w                                           # forward 3*E*w + backward 5*E*w
    step_roof = dict(alg_bytes=step_alg, GBps=round(step_alg / (ms_per_step * 1e-3) / 1e9, 1),
                     frac=round(step_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     sum_kernel_us=round(sum(kt.values()), 1), event_pad_us=round(event_pad, 2))

    eager_ms = eager_inline_ms = None
    if rank == 0 and world == 1 and not args.no_eager:
        eager_ms = round(eager_module_step(args.workload, device, args.dtype), 4)
        eager_inline_ms = round(eager_module_step(args.workload, device, args.dtype, engine_inline=True), 4)
        from mga_yolo_amd import handoff_report
        handoff_report()
    slice_res = None
    if rank == 0 and world == 1 and not args.no_eager:
        slice_res = slice_step(args.workload, device, dtype_name=args.dtype)
    # SURVEY 8d images/s definition (2): the train-step harness, under DDP when N > 1 (every rank takes part)
    harness_res = None
    if not args.no_harness and not args.no_eager:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import harness
            harness_res = harness.run(MODEL_SCALE.get(args.workload, "n"), batch=batch, size=1280 if args.workload == "cfg4" else 640,
                                      steps=args.harness_steps, warmup=3, device=device, world=world, rank=rank)
        except Exception as e:                                    # the harness is an extra key: its failure must not cost the headline line
            harness_res = dict(error=f"{type(e).__name__}: {e}"[:300])
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload, args.cpu_seconds)

    if rank == 0:
        line = dict(metric="images/s (MaskCBAM fwd+bwd step at P3/P4/P5, YOLOv8n 640x640)", value=round(value, 1), unit="images/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4), ms_per_step_device=round(ms_device, 4),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype=args.dtype, data="synthetic",
                    config=dict(workload=desc, batch_per_gpu=batch, global_batch=batch * world,
                                levels=[list(s) for s in plan.shapes], parallelism=f"dp{world}", backend=(args.backend if world > 1 else None),
                                launch="eager" if args.no_graph else ((f"hipGraph replay, {max(1, args.steps_per_graph)} step(s)/graph" if args.steps_per_graph > 1 else "hipGraph replay, 1 graph/step") if world == 1 else "hipGraph replay, 2 graphs/step (split at the gradient-exchange join)"),
                                grad_exchange=None if world == 1 else dict(
                                    collective=f"{'RCCL' if args.backend == 'nccl' else args.backend} all-reduce(mean), one side stream, buckets in order",
                                    block_bucket_bytes=plan.grad_bucket.numel() * 4, payload_mb=payload_mb, payload_buckets=[b.numel() * 4 for b in payload],
                                    total_bytes=exchange.nbytes(),
                                    placement="started after the backward that produces the gradients, overlapped with the next step's parameter-free k_pool, joined before the first kernel that reads a parameter (the DDP contract); the payload stands for the out-of-scope layers' gradients, whose own backward -- absent here -- is what hides most of it in a whole-model step (see train_harness)"),
                                handoff_kernels=bool(plan.fuse_forward)),
                    roofline=roofline, step_roofline=step_roof, kernels=kernels, cpu_baseline=cpu,
                    eager_module_ms_per_step=eager_ms, eager_module_ms_per_step_engine_inline=eager_inline_ms, layer_loop_slice=slice_res,
                    train_harness=harness_res,
                    eager_module_note="the unchanged reference layer loop's path: three MaskCBAM modules through autograd, one library call per level each way, launches issued from Python (no PyramidPlan / hipGraph), gradients reset to None every step as the trainer does; engine_inline = autograd engine on the calling thread (the blocks' marginal cost inside a whole-model backward, which pays the engine's worker-thread wake-up once for all layers)",
                    lib=_lib.load().mgacbam_build_info().decode())
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
