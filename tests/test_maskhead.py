"""MGAMaskHead: 1x1 conv -> BatchNorm2d -> SiLU -> 3x3 conv producing the mask logits that MaskCBAM consumes
(mga_yolo/nn/modules/segmentation.py:56-110; layer-loop hand-off mga_yolo/model/model.py:57-74).  Pinned by outputs of the
reference's own class (tests/golden/head_*.npz, head_checksums.json, written by oracle/gen_golden_head.py).
CPU: the oracle vs those goldens, the module mirror (state_dict, init, host path).  GPU: the HIP path (C ABI mgahead_*) vs the goldens."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, checksum, rel_err
from oracle import maskhead_oracle as HO

TOL = 1e-4


def head_golden_names():
    return sorted(f[len("head_"):-4] for f in os.listdir(GOLDEN) if f.startswith("head_") and f.endswith(".npz"))


def load_head_golden(name):
    z = np.load(os.path.join(GOLDEN, f"head_{name}.npz"), allow_pickle=False)
    return dict(x=torch.from_numpy(z["x"]), g=torch.from_numpy(z["g_logits"]),
                params={k[len("param."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param.")},
                out={k[len("out."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("out.")},
                meta=json.loads(bytes(z["meta"]).decode()))


GRAD_KEYS = ("gx", "gw1", "ggamma", "gbeta", "gwh", "gbh")


# Entries the REFERENCE's own fp32 arithmetic cannot pin to the usual bar.  dc_offset (features = N(0,1) + 50: per-channel mean^2 / var of
# z ~ 1e4, the case that exposes a cancelling BatchNorm variance): dW1 = sum g_z x with sum g_z = 0 and x ~ 50 is a difference of large
# terms; against a float64 evaluation of the same graph the reference's stored dW1 is 9.9e-5 off and a plain fp32 einsum 1.8e-4
# (measured when the golden was made), everything else 5e-6.  Its dW1 is therefore compared at 5e-4; all other entries at the bar.
LOOSE = {("dc_offset", "gw1"): 5e-4,
         # hid192 (C = 768: sums of 768 products per output): the ORACLE's fp32 einsum against the reference's fp32 conv depends on the
         # host's BLAS blocking -- 2e-5 holds on the build container's CPU, not on the GPU box's (EPYC 9575F); 1e-4 is the bar anyway
         **{("hid192", k): 1e-4 for k in ("logits", "gx", "gw1", "ggamma", "gbeta", "gwh", "gbh", "running_mean", "running_var")}}
_CASE = [None]


def _close(got, want, tol, what):
    # relative to the tensor's scale, with an absolute floor for all-zero expectations (stride probe: logits are exactly the bias)
    tol = max(tol, LOOSE.get((_CASE[0], what), 0.0))
    scale = float(want.double().abs().max())
    err = float((got.double() - want.double()).abs().max())
    assert err <= tol * max(scale, 1e-6), (what, err, scale)


@pytest.mark.parametrize("name", head_golden_names())
def test_oracle_matches_the_reference_golden(name):
    d = load_head_golden(name)
    _CASE[0] = name
    m = d["meta"]
    p = HO.HeadParams.from_state_dict(d["params"], eps=m["eps"], momentum=m["momentum"])
    logits, c = HO.forward(d["x"], p, training=m["training"])
    g = HO.backward(d["g"], d["x"], p, c, training=m["training"])
    _close(logits, d["out"]["logits"], 1e-5, "logits")
    _close(c.new_running_mean, d["out"]["running_mean"], 1e-5, "running_mean")
    _close(c.new_running_var, d["out"]["running_var"], 1e-5, "running_var")
    for k in GRAD_KEYS:
        _close(g[k].reshape(d["out"][k].shape), d["out"][k], 2e-5, k)
    y2, gx2 = HO.reference_form_step(d["x"], p, d["g"], training=m["training"])
    _close(y2, d["out"]["logits"], 1e-6, "eager-op form logits")
    _close(gx2, d["out"]["gx"], 1e-6, "eager-op form gx")


def _module_from_golden(d, device="cpu"):
    from mga_yolo_amd import MGAMaskHead
    hid, C = d["params"]["proj.0.weight"].shape[:2]
    m = MGAMaskHead(C, hid)
    m.load_state_dict(d["params"], strict=True)                  # a reference state_dict loads as is
    bn = m.proj[1]
    bn.eps, bn.momentum = d["meta"]["eps"], d["meta"]["momentum"]
    m.train(d["meta"]["training"])
    return m.to(device)


def _check_module(d, m, dev, tol):
    x = d["x"].to(dev).requires_grad_(True)
    y = m(x)
    y.backward(d["g"].to(dev))
    o = d["out"]
    _close(y.detach().cpu(), o["logits"], tol, "logits")
    _close(x.grad.cpu(), o["gx"], tol, "gx")
    p = dict(m.named_parameters())
    for k, name in (("gw1", "proj.0.weight"), ("ggamma", "proj.1.weight"), ("gbeta", "proj.1.bias"), ("gwh", "head.weight"), ("gbh", "head.bias")):
        _close(p[name].grad.cpu(), o[k], tol, k)
    sd = m.state_dict()
    _close(sd["proj.1.running_mean"].cpu(), o["running_mean"], tol, "running_mean")
    _close(sd["proj.1.running_var"].cpu(), o["running_var"], tol, "running_var")
    assert int(sd["proj.1.num_batches_tracked"]) == int(o["num_batches_tracked"])


@pytest.mark.parametrize("name", head_golden_names())
def test_module_host_path_matches_the_reference_golden(name):
    d = load_head_golden(name)
    _CASE[0] = name
    _check_module(d, _module_from_golden(d), "cpu", 2e-5)


def test_module_mirror_contract():
    """Same constructor / state_dict / initialisation order as the reference class (segmentation.py:56-104): the same seed gives the
    same initial values (recorded in the p3_like golden, which was built at seed 0 without perturbation)."""
    from mga_yolo_amd import MGAMaskHead
    d = load_head_golden("p3_like")
    torch.manual_seed(0)
    m = MGAMaskHead(64, 16)
    sd = m.state_dict()
    assert list(sd) == list(d["params"])
    for k, v in d["params"].items():
        assert torch.equal(sd[k], v), k
    assert m.cfg.in_channels == 64 and m.cfg.hidden_channels == 16 and m.cfg.out_channels == 1 and m.cfg.norm == "bn"
    assert "in=64, hidden=16, out=1, norm=bn, act=SiLU, dropout=0.0" in repr(m)
    assert m.hip_path()
    import copy, pickle
    assert torch.equal(copy.deepcopy(m).head.weight, m.head.weight) and pickle.loads(pickle.dumps(m)).cfg.hidden_channels == 16
    # constructor variants keep the reference's structure (served by torch ops)
    v = MGAMaskHead(32, 8, out_channels=2, norm="ln", act=torch.nn.ReLU, dropout=0.1)
    assert not v.hip_path() and list(v.state_dict()) == ["proj.0.weight", "proj.1.ln.weight", "proj.1.ln.bias", "head.weight", "head.bias"]
    assert v(torch.randn(1, 32, 5, 5)).shape == (1, 2, 5, 5)
    assert MGAMaskHead(8, 8, norm=None, act=None)(torch.randn(1, 8, 3, 3)).shape == (1, 1, 3, 3)


# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", head_golden_names())
def test_device_path_matches_the_reference_golden(built_lib, name):
    """HIP path (C ABI mgahead_forward / mgahead_backward through the module) vs what the reference's own MGAMaskHead produced:
    logits, every gradient, BatchNorm running statistics and num_batches_tracked."""
    d = load_head_golden(name)
    _CASE[0] = name
    m = _module_from_golden(d, "cuda")
    assert m.hip_path()
    _check_module(d, m, "cuda", TOL)
    assert "libmgacbam.so" in open("/proc/self/maps").read()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg2_p3", "cfg2_p4", "cfg2_p5", "cfg3_p3", "cfg3_p4", "cfg3_p5",
                                  "cfg5_640_p3", "cfg5_640_p4", "cfg5_640_p5", "cfg5_1280_p3", "cfg4_p3_192"])
def test_device_full_size_checksums_vs_reference(built_lib, name):
    """BASELINE configs[1], [2] and the mask-head leg of configs[3] / [4] (YOLOv8l widths C 256 / 512 / 512, hidden 64 / 128 / 128, 640 and
    1280 px; the reference evaluated in fp32 on bf16-rounded inputs, the device path fed bf16 tensors)."""
    from mga_yolo_amd import MGAMaskHead
    ref = json.load(open(os.path.join(GOLDEN, "head_checksums.json")))["big"][name]
    B, C, hid, H, W = ref["shape"]
    bf16 = ref.get("recipe") == "bf16"
    tol = 1e-3 if bf16 else TOL                                    # (the bar of the bf16 MaskCBAM checksums, test_gpu_parity.py)
    torch.manual_seed(0)
    m = MGAMaskHead(C, hid)
    m.proj[1].eps, m.proj[1].momentum = ref["eps"], ref["momentum"]
    m.cuda().train()
    g = torch.Generator().manual_seed(1234)                        # the recipe of oracle/gen_golden_head.py:data()
    x = torch.randn(B, C, H, W, generator=g)
    gl = torch.randn(B, 1, H, W, generator=g)
    if bf16:
        x, gl = x.bfloat16(), gl.bfloat16()
    xd = x.cuda().requires_grad_(True)
    y = m(xd)
    assert y.dtype == x.dtype
    y.backward(gl.cuda())
    p = dict(m.named_parameters())
    got = dict(logits=y.detach(), gx=xd.grad, gw1=p["proj.0.weight"].grad, ggamma=p["proj.1.weight"].grad, gbeta=p["proj.1.bias"].grad,
               gwh=p["head.weight"].grad, gbh=p["head.bias"].grad, running_mean=m.proj[1].running_mean, running_var=m.proj[1].running_var)
    report = []
    for k, v in got.items():
        c = checksum(v.float())
        scale = ref[k]["abs"] + 1e-12
        for f in ("sum", "wsum", "abs"):
            if not abs(c[f] - ref[k][f]) <= tol * scale:
                report.append(f"{k}.{f}: got {c[f]:.6f} want {ref[k][f]:.6f}")
    assert not report, f"{name}: " + "; ".join(report)
    # run-to-run reproducibility (fixed-order sums, no float atomics)
    m.zero_grad(); xd.grad = None
    y2 = m(xd); y2.backward(gl.cuda())
    assert torch.equal(y2, y) and torch.equal(xd.grad, got["gx"]) and torch.equal(p["proj.0.weight"].grad, got["gw1"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,hid,H,W", [(2, 16, 64, 3, 100),     # hidden > 32 takes 1 pixel per thread; a run of 256 cannot hold W + 1 either side: 2
                                         (1, 8, 8, 2, 333),       # H*W % 4 != 0: the scalar-access form, 4 pixels per thread
                                         (1, 8, 40, 1, 500),      # the widest row k_head_out takes: 16 outputs per run of 1024
                                         (1, 16, 200, 5, 36)])    # more hidden channels than one LDS chunk of constants (128)
def test_device_wide_rows_vs_live_oracle(built_lib, B, C, hid, H, W):
    """k_head_out holds a pixel run and W + 1 pixels either side in 256 x {1, 2, 4} staged pixels: rows that force the larger forms, the
    4-byte access form and the widest row, against the oracle evaluated live (fp32, 1e-4 as for the goldens)."""
    from mga_yolo_amd import MGAMaskHead
    torch.manual_seed(11)
    m = MGAMaskHead(C, hid).train()
    g = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(0.3 * torch.randn(p_.shape, generator=g))
    x = torch.randn(B, C, H, W, generator=g)
    gl = torch.randn(B, 1, H, W, generator=g)
    p = HO.HeadParams.from_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    lo, c = HO.forward(x, p, True)
    go = HO.backward(gl, x, p, c, True)
    m.cuda()
    xd = x.cuda().requires_grad_(True)
    y = m(xd)
    y.backward(gl.cuda())
    assert rel_err(y, lo) < TOL and rel_err(xd.grad, go["gx"]) < TOL
    assert rel_err(m.head.weight.grad, go["gwh"]) < TOL and rel_err(m.proj[0].weight.grad.reshape(hid, C), go["gw1"]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_device_half_precision_io(built_lib, dtype, tol):
    """fp16 / bf16 features and logits with fp32 accumulation and fp32 parameters, against the fp32 oracle on the rounded inputs."""
    from mga_yolo_amd import MGAMaskHead
    torch.manual_seed(3)
    m = MGAMaskHead(128, 32).cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 128, 20, 20, generator=g).to(dtype)
    gl = torch.randn(4, 1, 20, 20, generator=g).to(dtype)
    p = HO.HeadParams.from_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    lo, c = HO.forward(x.float(), p, True)
    go = HO.backward(gl.float(), x.float(), p, c, True)
    xd = x.cuda().requires_grad_(True)
    y = m(xd)
    assert y.dtype == dtype
    y.backward(gl.cuda())
    assert xd.grad.dtype == dtype
    assert rel_err(y.float(), lo) < tol and rel_err(xd.grad.float(), go["gx"]) < tol
    assert rel_err(m.proj[0].weight.grad.reshape(32, 128), go["gw1"]) < tol and rel_err(m.head.weight.grad, go["gwh"]) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_device_halved_module_in_eval_mode(built_lib, dtype, tol):
    """The validator calls model.half() on the EMA copy whenever AMP is on (U/engine/validator.py:147-149; half=True predict does the
    same): every parameter AND the BatchNorm running statistics are half precision then.  The HIP path must serve that module
    (fp32 copies of the statistics: eval mode writes nothing back) and agree with the fp32 oracle on the rounded state."""
    from mga_yolo_amd import MGAMaskHead
    torch.manual_seed(5)
    m = MGAMaskHead(128, 32)
    with torch.no_grad():
        m.proj[1].running_mean.normal_(0.0, 0.3)
        m.proj[1].running_var.uniform_(0.5, 1.5)
    m = m.cuda().to(dtype).eval()
    assert m.hip_path() and m.proj[1].running_mean.dtype == dtype
    x = torch.randn(3, 128, 20, 20, generator=torch.Generator().manual_seed(8)).to(dtype)
    with torch.no_grad():
        y = m(x.cuda())
    assert y.dtype == dtype and m.proj[1].running_mean.dtype == dtype
    p = HO.HeadParams.from_state_dict({k: v.detach().float().cpu() for k, v in m.state_dict().items()})
    lo, _ = HO.forward(x.float(), p, False)
    assert rel_err(y.float(), lo) < tol
    # a TRAINING module with half-precision buffers (not something the reference trainer produces) still works: the containers' torch ops
    m.train()
    assert m(x.cuda()).shape == y.shape


@pytest.mark.gpu
def test_device_pyramid_call_and_layer_loop_hand_off(built_lib):
    """The three heads in ONE library call each way == three module calls; and the layer-loop hand-off of the reference
    (mga_yolo/model/model.py:57-74): head -> [feat, mask] -> MaskCBAM, one backward through both, against the oracles."""
    from mga_yolo_amd import MGAMaskHead, MaskCBAM, mask_head_pyramid
    from oracle import maskcbam_oracle as CO
    shapes = [(4, 64, 16, 16, 16), (4, 128, 8, 8, 32), (4, 256, 4, 4, 64)]
    mods, xs = [], []
    for i, (B, C, H, W, hid) in enumerate(shapes):
        torch.manual_seed(i)
        mods.append(MGAMaskHead(C, hid).cuda().train())
        xs.append(torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(40 + i)).cuda())
    single = [m(x) for m, x in zip(mods, xs)]
    for m in mods:                                                # the second call must start from the same running statistics
        m.proj[1].reset_running_stats()
    lv = []
    for m, x in zip(mods, xs):
        bn = m.proj[1]
        lv.append((x, m.proj[0].weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, m.head.weight,
                   m.head.bias, bn.eps, bn.momentum, True))
    for a, b in zip(mask_head_pyramid(lv), single):
        assert torch.equal(a, b)
    # hand-off: logits -> MaskCBAM mask input; gradient flows back through the block into the head
    B, C, H, W, hid = shapes[0]
    torch.manual_seed(0)
    head, blk = MGAMaskHead(C, hid).cuda().train(), MaskCBAM(C).cuda()
    x = xs[0].clone().requires_grad_(True)
    gy = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(9)).cuda()
    y = blk([x, head(x)])
    y.backward(gy)
    hp = HO.HeadParams.from_state_dict({k: v.detach().cpu() for k, v in head.state_dict().items()})
    hp.running_mean.zero_(); hp.running_var.fill_(1.0)           # (the state_dict was read after the forward updated them)
    xc = x.detach().cpu()
    logits, hc = HO.forward(xc, hp, True)
    cp = CO.Params.from_state_dict({k: v.detach().cpu() for k, v in blk.state_dict().items()})
    y_o, cc = CO.forward(xc, logits, cp)
    g_o = CO.backward(gy.cpu(), xc, logits, cp, CO.Config(), cc)
    gh = HO.backward(g_o["gmask"], xc, hp, hc, True)
    assert rel_err(y, y_o) < TOL
    assert rel_err(x.grad, g_o["gx"] + gh["gx"]) < TOL             # the feature feeds both the block and its mask head
    assert rel_err(head.proj[0].weight.grad.reshape(hid, C), gh["gw1"]) < TOL and rel_err(head.head.weight.grad, gh["gwh"]) < TOL
