"""Segmentation loss measurement on an MI355X: one forward + backward of the multi-scale loss on the mask logits of
BASELINE configs[1] (32 x 1 x {80,40,20}^2, fp32, default BCE + Dice mode), through the C ABI (2 + 1 launches for all levels),
replayed from a hipGraph; beside the same step with torch's own device ops (what the reference's module would launch on a GPU:
~40 kernels) and the oracle on the host cores.  The tensors total 1.1 MB, so this is a launch-latency measurement, not a
bandwidth one: algorithmic bytes (logits + target read twice, gradient written once) are reported for completeness.
Not the headline benchmark (bench.py is); prints one JSON line."""
import ctypes as C
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mga_yolo_amd import _lib
from mga_yolo_amd.segloss import SegLossConfig, SegmentationLoss

B, SIZES = 32, [(80, 80), (40, 40), (20, 20)]
g = torch.Generator().manual_seed(1234)
logits = [torch.randn(B, 1, h, w, generator=g).cuda() for h, w in SIZES]
targets = [(torch.rand(B, 1, h, w, generator=g) > 0.9).float().cuda() for h, w in SIZES]
grads = [torch.empty_like(x) for x in logits]
lib = _lib.load()
n = len(SIZES)
lv = (_lib.SegLevel * n)()
for l, (x, t, gx) in enumerate(zip(logits, targets, grads)):
    lv[l].logits, lv[l].target, lv[l].glogits = x.data_ptr(), t.data_ptr(), gx.data_ptr()
    lv[l].B, lv[l].H, lv[l].W, lv[l].Ht, lv[l].Wt, lv[l].dtype, lv[l].scale_weight = B, x.shape[2], x.shape[3], x.shape[2], x.shape[3], _lib.F32, 1.0
cfg = _lib.SegCfg(1.0, 1.0, 1.0, 1.0)
ws = torch.empty(lib.mgaseg_ws_bytes(lv, n), dtype=torch.uint8, device="cuda")
out = torch.empty(1 + 3 * n, device="cuda")
gout = torch.ones(1, device="cuda")


def step():
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mgaseg_forward(lv, n, C.byref(cfg), ws.data_ptr(), ws.numel(), out.data_ptr(), st), "fwd")
    _lib.check(lib.mgaseg_backward(lv, n, C.byref(cfg), ws.data_ptr(), ws.numel(), gout.data_ptr(), st), "bwd")


def timed(fn, steps=500, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph): step()
us_graph = timed(graph.replay)
us_eager = timed(step)

# the same step with torch's device ops (the reference module's op sequence on a GPU)
ref = SegmentationLoss(SegLossConfig())
xs = [x.clone().requires_grad_(True) for x in logits]
def torch_step():
    for x in xs: x.grad = None
    total, _ = ref._torch_forward([(k, x, t, 1.0) for k, x, t in zip(("p3", "p4", "p5"), xs, targets)])
    total.backward()
us_torch = timed(torch_step, steps=100, warm=10)

elems = sum(B * h * w for h, w in SIZES)
res = dict(block="SegmentationLoss", unit="us/step", us_per_step=round(us_graph, 2), us_per_step_eager_abi=round(us_eager, 2),
           us_per_step_torch_device_ops=round(us_torch, 2), launches_per_step=3, alg_bytes=elems * 4 * 5,
           GBps=round(elems * 4 * 5 / us_graph / 1e3, 1), images_per_s=round(B / us_graph * 1e6, 1))
if "--cpu" in sys.argv:
    from oracle import segloss_oracle as O
    torch.set_num_threads(min(16, os.cpu_count()))
    cp = {k: x.cpu().clone().requires_grad_(True) for k, x in zip(("p3", "p4", "p5"), logits)}
    ct = [t.cpu() for t in targets]
    def one():
        for v in cp.values(): v.grad = None
        O.forward(cp, ct, O.SegLossConfig())[0].backward()
    one(); t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 5: one(); k += 1
    res["cpu_baseline"] = dict(us_per_step=round((time.perf_counter() - t0) / k * 1e6, 1), cores=min(16, os.cpu_count()), kind="port")
print(json.dumps(res))
