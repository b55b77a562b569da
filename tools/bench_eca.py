"""MaskECA measurement on an MI355X: images/s of a forward+backward step at P3/P4/P5 (YOLOv8n widths, batch 32,
640x640, fp32), per-kernel durations in step order and achieved GB/s against the block's algorithmic bytes (forward
3*E*4 B: k_eca_pool 1, k_eca_apply 2; backward 5*E*4 B: k_eca_reduce 2, k_eca_bwd 3), beside the oracle's eager-op form on the
host cores.  Not the headline benchmark (bench.py is); prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mga_yolo_amd import MaskECA
from mga_yolo_amd.plan import EcaPyramidPlan

LV = [(64, 80, 80), (128, 40, 40), (256, 20, 20)]
B = 32
shapes, params, cfgs = [], [], []
for C, H, W in LV:
    torch.manual_seed(0); m = MaskECA(C)
    shapes.append((B, C, H, W)); params.append((m.conv1d.weight, m.beta)); cfgs.append(m.eca_config())
plan = EcaPyramidPlan(shapes, params, cfgs)
g = torch.Generator().manual_seed(1234)
for l, s in enumerate(shapes):
    plan.x[l].copy_(torch.nn.functional.silu(torch.randn(*s, generator=g)))
    plan.mask[l].copy_(torch.randn(s[0], 1, s[2], s[3], generator=g) - 2.0)
    plan.gy[l].copy_(torch.randn(*s, generator=g))
graph = plan.capture(lambda: (plan.forward(), plan.backward()))
steps = 200
for _ in range(20): graph.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): graph.replay()
torch.cuda.synchronize(); el = time.perf_counter() - t0
E = plan.elements()
out = dict(block="MaskECA", value=round(B * steps / el, 1), unit="images/s", ms_per_step=round(el * 1e3 / steps, 4),
           step_alg_bytes=8 * E * 4, step_GBps=round(8 * E * 4 / (el / steps) / 1e9, 1), launches_per_step=4)
if "--cpu" in sys.argv:
    from oracle import maskeca_oracle as Eo
    torch.set_num_threads(min(16, os.cpu_count()))
    data = []
    for (C, H, W) in LV:
        data.append((torch.randn(B, C, H, W), torch.randn(B, 1, H, W) - 2, torch.randn(B, C, H, W), Eo.EcaParams.default_init(C)))
    one = lambda: [Eo.reference_form_step(x, mk, p, Eo.EcaConfig(), gy) for x, mk, gy, p in data]
    one(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 8: one(); n += 1
    out["cpu_baseline"] = dict(value=round(B * n / (time.perf_counter() - t0), 2), unit="images/s", cores=min(16, os.cpu_count()), kind="port")
print(json.dumps(out))
