"""GPU parity tests (-m gpu): the HIP path behind the C ABI vs the oracle, the golden vectors from the reference, and
size-independent properties at BASELINE.json's full sizes.  Tolerance: 1e-4 relative (fp32), the bar north_star states;
index outputs (arg-max positions / channels, nearest-resize) must be bit-exact."""
import os

import pytest
import torch

from conftest import checksum, elem_err, golden_case_names, load_golden, rel_err, synth
from oracle import maskcbam_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star: within 1e-4 relative fp32
GRADS = ("gx", "gmask", "gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta")


@pytest.fixture(scope="module")
def F():
    import mga_yolo_amd.functional as Fn
    from mga_yolo_amd import _lib
    _lib.load()                      # fail loudly if libmgacbam.so is missing
    return Fn


def _cfg(F, d):
    m = d["meta"]
    hidden = d["params"]["cam_mlp.0.weight"].shape[0]
    return F.BlockConfig(hidden=hidden, k=m["k"], use_sigmoid_mask=m["use_sigmoid_mask"], tiny_thr=m["tiny_thr"], eps=m["eps"])


def _params_dev(d, requires_grad=False):
    keys = ("cam_mlp.0.weight", "cam_mlp.0.bias", "cam_mlp.2.weight", "cam_mlp.2.bias", "sam_conv.weight", "beta")
    return [d["params"][k].cuda().requires_grad_(requires_grad) for k in keys]


def _run_gpu(F, d):
    x = d["x"].cuda().requires_grad_(True)
    mask = None if d["mask"] is None else d["mask"].cuda().requires_grad_(True)
    ps = _params_dev(d, True)
    y = F.mask_cbam(x, mask, *ps, _cfg(F, d))
    y.backward(d["gy"].cuda())
    g = dict(gx=x.grad, gmask=None if mask is None else mask.grad, gw1=ps[0].grad, gb1=ps[1].grad, gw2=ps[2].grad,
             gb2=ps[3].grad, gwsa=ps[4].grad, gbeta=ps[5].grad)
    return y.detach(), g


@pytest.mark.parametrize("name", golden_case_names())
def test_forward_stages_vs_oracle(F, name):
    """Every statistic the forward kernels save, compared with the oracle's intermediates: localises a failure to a kernel."""
    d = load_golden(name)
    p = O.Params.from_state_dict(d["params"])
    ocfg = O.Config(use_sigmoid_mask=d["meta"]["use_sigmoid_mask"], tiny_thr=d["meta"]["tiny_thr"], eps=d["meta"]["eps"])
    y_o, c = O.forward(d["x"], d["mask"], p, ocfg)
    y, v = F.forward_with_ctx(d["x"].cuda(), None if d["mask"] is None else d["mask"].cuda(), _params_dev(d), _cfg(F, d))
    torch.cuda.synchronize()
    B, C, H, W = d["x"].shape
    report = []

    def close(name_, got, want, tol=TOL):
        e = rel_err(got, want)
        if not e < tol:
            report.append(f"{name_}: rel_err={e:.3e}")

    def same(name_, got, want):
        if not torch.equal(got.cpu().long(), want.long()):
            report.append(f"{name_}: {int((got.cpu().long() != want.long()).sum())} mismatching indices")

    if d["mask"] is not None:
        close("S", v["S"], c.S); close("use", v["use"], c.use); close("den", v["den"], c.den)
        close("mavg", v["mavg"], c.mavg)
    close("avg", v["avg"], c.avg); close("mx", v["mx"], c.mx)
    same("valid", v["valid"], c.valid.long())
    same("amax[valid]", v["amax"].cpu() * c.valid.long(), c.amax * c.valid.long())
    close("h_avg", v["h_avg"], c.h_avg); close("h_mx", v["h_mx"], c.h_mx); close("ca", v["ca"], c.ca)
    close("planes", v["planes"], c.planes.reshape(B, 3, H * W))
    if name not in ("stride_probe",):            # all-zero input: every channel ties; first index still required
        pass
    same("cidx", v["cidx"], c.cidx)
    close("sa", v["sa"], c.sa)
    if "proj" in v and d["mask"] is not None:    # W1-projection planes saved for the backward (hidden <= 8)
        close("proj", v["proj"], torch.einsum("jc,bcn->bjn", p.w1, d["x"].reshape(B, C, H * W)))
    close("y", y, y_o)
    close("y_vs_reference", y, d["out"]["y"])
    assert not report, f"{name}: " + "; ".join(report)


@pytest.mark.parametrize("name", golden_case_names())
def test_fwd_bwd_vs_reference_golden(F, name):
    d = load_golden(name)
    y, g = _run_gpu(F, d)
    torch.cuda.synchronize()
    report = []
    e = rel_err(y, d["out"]["y"])
    if not e < TOL:
        report.append(f"y {e:.3e}")
    for k in GRADS:
        if k == "gmask" and d["mask"] is None:
            assert g[k] is None
            continue
        assert g[k].shape == d["out"][k].shape, k
        e = rel_err(g[k], d["out"][k])
        if not e < TOL:
            report.append(f"{k} {e:.3e}")
        # element-wise as well: every element of at least 1e-3 of the tensor's scale within 1e-3 of ITS OWN value (smaller ones
        # within 1e-6 of the scale) -- the tensor-scale bound above says nothing about small elements
        ee = elem_err(g[k], d["out"][k])
        if not ee < 1e-3:
            report.append(f"{k} element-wise {ee:.3e}")
    if not elem_err(y, d["out"]["y"]) < 1e-3:
        report.append(f"y element-wise {elem_err(y, d['out']['y']):.3e}")
    assert not report, f"{name}: " + "; ".join(report)


@pytest.mark.parametrize("name", ["cfg1_p3", "cfg1_p4", "cfg1_p5", "cfg2_p3", "cfg2_p4", "cfg2_p5", "survey_A1", "survey_nomask"])
def test_full_size_checksums_vs_reference(F, checksums, name):
    """BASELINE.json configs 1 and 2 (YOLOv8n, B=2 and B=32, 640x640): checksums of y and every gradient against the
    values the reference itself produced (tests/golden/checksums.json)."""
    ref = checksums["big"][name]
    B, C, H, W = ref["shape"]
    x, mask, gy = synth(B, C, H, W, mask_kind=ref["mask_kind"])
    if ref["recipe"] == "ysum":
        gy = torch.ones_like(x)
    p = O.Params.default_init(C)
    d = dict(x=x, mask=mask, gy=gy, params={"cam_mlp.0.weight": p.w1, "cam_mlp.0.bias": p.b1, "cam_mlp.2.weight": p.w2,
                                            "cam_mlp.2.bias": p.b2, "sam_conv.weight": p.wsa, "beta": p.beta},
             meta=dict(k=7, use_sigmoid_mask=True, tiny_thr=1e-4, eps=1e-6))
    y, g = _run_gpu(F, d)
    got = dict(y=y, **{k: v for k, v in g.items() if v is not None})
    report = []
    for k, v in got.items():
        c = checksum(v)
        scale = ref[k]["abs"] + 1e-12
        for f in ("sum", "wsum", "abs"):
            if not abs(c[f] - ref[k][f]) <= TOL * scale:
                report.append(f"{k}.{f}: got {c[f]:.6f} want {ref[k][f]:.6f}")
    assert not report, f"{name}: " + "; ".join(report)


@pytest.mark.parametrize("shape,mask_kind", [((32, 64, 80, 80), "sparse"), ((32, 128, 40, 40), "randn"), ((32, 256, 20, 20), "sparse"),
                                             ((4, 192, 40, 40), "mixed"), ((2, 384, 20, 20), "randn"), ((3, 48, 17, 17), "mixed"),
                                             ((1, 256, 160, 160), "sparse"), ((2, 512, 40, 40), "randn"), ((1, 768, 20, 20), "mixed"),
                                             ((5, 64, 24, 40), "mixed"), ((9, 32, 6, 10), "randn"), ((1, 1, 1, 1), "randn"),
                                             ((2, 3, 2, 3), "randn"), ((1, 16, 1, 37), "sparse"), ((11, 24, 13, 4), "mixed"),
                                             # BASELINE configs[2] (YOLOv8s, 32 images per GPU) at FULL batch and configs[3]'s P4 (8 x 512 x 80 x 80):
                                             # twice config 2's workgroups per launch; a wrong tile or halo shows here element by element
                                             ((32, 128, 80, 80), "sparse"), ((32, 256, 40, 40), "randn"), ((32, 512, 20, 20), "sparse"),
                                             ((8, 512, 80, 80), "sparse")])
def test_full_size_vs_oracle_live(F, shape, mask_kind):
    """Same seeded inputs through the oracle (CPU) and the HIP path, element-wise, at config-2 / config-3 sizes and the odd shapes."""
    B, C, H, W = shape
    if mask_kind == "mixed" and B < 2:
        mask_kind = "randn"
    x, mask, gy = synth(B, C, H, W, seed=77, mask_kind=mask_kind)
    p = O.Params.default_init(C, seed=3)
    p.beta.fill_(0.3)
    y_o, ctx = O.forward(x, mask, p)
    g_o = O.backward(gy, x, mask, p, O.Config(), ctx)
    d = dict(x=x, mask=mask, gy=gy, params={"cam_mlp.0.weight": p.w1, "cam_mlp.0.bias": p.b1, "cam_mlp.2.weight": p.w2,
                                            "cam_mlp.2.bias": p.b2, "sam_conv.weight": p.wsa, "beta": p.beta},
             meta=dict(k=7, use_sigmoid_mask=True, tiny_thr=1e-4, eps=1e-6))
    y, g = _run_gpu(F, d)
    report = []
    if not rel_err(y, y_o) < TOL:
        report.append(f"y {rel_err(y, y_o):.3e}")
    for k in GRADS:
        if g_o[k] is None:
            continue
        e = rel_err(g[k], g_o[k])
        if not e < TOL:
            report.append(f"{k} {e:.3e}")
    # every ELEMENT of the feature-sized outputs (conftest.elem_err: own-value relative with a floor of 1e-3 of the scale)
    for k, got, want in (("y", y, y_o), ("gx", g["gx"], g_o["gx"]), ("gmask", g["gmask"], g_o["gmask"])):
        if want is not None and not elem_err(got, want) < 1e-3:
            report.append(f"{k} element-wise {elem_err(got, want):.3e}")
    assert not report, "; ".join(report)


def test_pyramid_call_equals_per_level_calls(F):
    shapes = [(4, 64, 40, 40), (4, 128, 20, 20), (4, 256, 10, 10)]
    lv, single = [], []
    for i, (B, C, H, W) in enumerate(shapes):
        x, mask, gy = synth(B, C, H, W, seed=10 + i)
        p = O.Params.default_init(C, seed=i)
        ps = [t.cuda() for t in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
        cfg = F.BlockConfig(hidden=p.w1.shape[0])
        lv.append((x.cuda(), mask.cuda(), ps, cfg))
        single.append(F.mask_cbam(x.cuda(), mask.cuda(), *ps, cfg))
    outs = F.mask_cbam_pyramid(lv)
    for a, b in zip(outs, single):
        assert torch.equal(a, b)


def test_bitwise_reproducible_and_batch_independent(F):
    """Two-stage reductions, no float atomics: identical bits run to run; a sample's result does not depend on its batch."""
    B, C, H, W = 8, 64, 40, 40
    x, mask, gy = synth(B, C, H, W, seed=5, mask_kind="mixed")
    p = O.Params.default_init(C)
    base = dict(x=x, mask=mask, gy=gy, params={"cam_mlp.0.weight": p.w1, "cam_mlp.0.bias": p.b1, "cam_mlp.2.weight": p.w2,
                                               "cam_mlp.2.bias": p.b2, "sam_conv.weight": p.wsa, "beta": p.beta},
                meta=dict(k=7, use_sigmoid_mask=True, tiny_thr=1e-4, eps=1e-6))
    y1, g1 = _run_gpu(F, base)
    y2, g2 = _run_gpu(F, base)
    assert torch.equal(y1, y2) and all(torch.equal(g1[k], g2[k]) for k in GRADS)
    sub = dict(base, x=x[2:3], mask=mask[2:3], gy=gy[2:3])
    ys, gs = _run_gpu(F, sub)
    assert rel_err(ys, y1[2:3]) < 1e-6 and rel_err(gs["gx"], g1["gx"][2:3]) < 1e-6 and rel_err(gs["gmask"], g1["gmask"][2:3]) < 1e-6


def test_backward_is_linear_in_gy(F):
    """Property at config-2 size: gradients are linear in the upstream gradient (no oracle needed)."""
    B, C, H, W = 32, 64, 80, 80
    x, mask, gy = synth(B, C, H, W, seed=9, mask_kind="sparse")
    p = O.Params.default_init(C)
    mk = lambda g_: dict(x=x, mask=mask, gy=g_, params={"cam_mlp.0.weight": p.w1, "cam_mlp.0.bias": p.b1, "cam_mlp.2.weight": p.w2,
                                                         "cam_mlp.2.bias": p.b2, "sam_conv.weight": p.wsa, "beta": p.beta},
                         meta=dict(k=7, use_sigmoid_mask=True, tiny_thr=1e-4, eps=1e-6))
    _, g1 = _run_gpu(F, mk(gy))
    _, g2 = _run_gpu(F, mk(-2.5 * gy))
    for k in GRADS:
        assert rel_err(g2[k], -2.5 * g1[k]) < 1e-5, k


@pytest.mark.parametrize("env", [dict(MGACBAM_POOL_TX="16", MGACBAM_POOL_CPT="1", MGACBAM_CHAN_TX="16"),
                                 dict(MGACBAM_POOL_TX="64", MGACBAM_POOL_CPT="4", MGACBAM_CHAN_TX="64"),
                                 dict(MGACBAM_POOL_TX="128", MGACBAM_POOL_CPT="2", MGACBAM_CHAN_TX="32"),
                                 dict(MGACBAM_POOL_TX="256", MGACBAM_POOL_CPT="4", MGACBAM_CHAN_TX="8"),
                                 dict(MGACBAM_POOL_TX="1", MGACBAM_POOL_CPT="1", MGACBAM_CHAN_TX="1")])
def test_every_launch_geometry_gives_the_same_answer(F, env, monkeypatch):
    """The launch-geometry hooks (rows x lanes, channels per thread) must not change results beyond rounding."""
    from mga_yolo_amd import _lib
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _lib.reload_env()                            # the library reads its knobs once; tell it the environment changed
    try:
        for name in ("base", "mixed_batch", "odd17_c48", "hidden1_nonsq", "nomask"):
            d = load_golden(name)
            y, g = _run_gpu(F, d)
            assert rel_err(y, d["out"]["y"]) < TOL, name
            for k in GRADS:
                if g[k] is not None:
                    assert rel_err(g[k], d["out"][k]) < TOL, (name, k)
    finally:
        monkeypatch.undo()
        _lib.reload_env()


@pytest.mark.parametrize("hw", [(20, 20), (6, 6), (5, 7), (40, 40)])     # 16-byte (8-element), 8-byte and scalar access paths
@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_half_precision_io(F, dtype, tol, hw):
    """fp16 / bf16 features and gradients with fp32 accumulation, against the fp32 oracle on the rounded inputs."""
    B, C, (H, W) = 4, 64, hw
    x, mask, gy = synth(B, C, H, W, seed=21)
    x, gy = x.to(dtype).float(), gy.to(dtype).float()
    p = O.Params.default_init(C)
    y_o, ctx = O.forward(x, mask, p)
    g_o = O.backward(gy, x, mask, p, O.Config(), ctx)
    xd = x.cuda().to(dtype).requires_grad_(True)
    md = mask.cuda().requires_grad_(True)
    ps = [t.cuda().requires_grad_(True) for t in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
    y = F.mask_cbam(xd, md, *ps, F.BlockConfig(hidden=p.w1.shape[0]))
    assert y.dtype == dtype
    y.backward(gy.cuda().to(dtype))
    assert xd.grad.dtype == dtype and md.grad.dtype == torch.float32
    assert rel_err(y.float(), y_o) < tol
    assert rel_err(xd.grad.float(), g_o["gx"]) < tol
    assert rel_err(md.grad, g_o["gmask"]) < tol
    assert rel_err(ps[0].grad, g_o["gw1"]) < tol and rel_err(ps[5].grad, g_o["gbeta"]) < tol


@pytest.mark.parametrize("out_hw,in_hw", [((80, 80), (640, 640)), ((40, 40), (640, 640)), ((20, 20), (640, 640)),
                                          ((68, 68), (544, 544)), ((17, 17), (544, 544)), ((7, 13), (10, 29)), ((33, 5), (100, 3))])
def test_nearest_resize_bit_exact(F, out_hw, in_hw):
    import torch.nn.functional as Fnn
    g = torch.Generator().manual_seed(3)
    src = (torch.rand(2, 1, *in_hw, generator=g) > 0.5).float() + torch.rand(2, 1, *in_hw, generator=g)
    want = Fnn.interpolate(src, size=out_hw, mode="nearest")
    got = F.resize_nearest(src.cuda(), *out_hw).cpu()
    assert torch.equal(got, want)
    assert torch.equal(got, O.nearest_resize(src, *out_hw))


def test_module_on_device_matches_module_on_host():
    from mga_yolo_amd import MaskCBAM
    torch.manual_seed(0)
    m = MaskCBAM(64)
    x, mask, gy = synth(2, 64, 20, 20)
    xh, mh = x.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    yh = m([xh, mh]); yh.backward(gy)
    gh = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad(); m.cuda()
    xd, md = x.cuda().requires_grad_(True), mask.cuda().requires_grad_(True)
    yd = m([xd, md]); yd.backward(gy.cuda())
    assert rel_err(yd, yh) < TOL and rel_err(xd.grad, xh.grad) < TOL and rel_err(md.grad, mh.grad) < TOL
    for n, p in m.named_parameters():
        assert rel_err(p.grad, gh[n]) < TOL, n
    # plain tensor input = no mask; forward hooks see a plain tensor
    seen = []
    h = m.register_forward_hook(lambda mod, inp, out: seen.append(type(out)))
    out = m(x.cuda())
    h.remove()
    assert seen == [torch.Tensor] and out.shape == x.shape


def test_native_library_is_loaded_in_this_process():
    """The round-end check records which .so files the test process loaded: make sure ours is one of them."""
    from mga_yolo_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libmgacbam.so" in maps


@pytest.mark.parametrize("want_gmask,use_proj", [(True, True), (True, False), (False, False)])
def test_plan_executor_matches_oracle_with_and_without_projection_planes(F, want_gmask, use_proj):
    """PyramidPlan (the graph-capturable executor): the backward that takes dL/dmask from the saved W1-projection planes
    (no x read in k_bwd_apply) and the one that re-reads x must both match the oracle; levels with hidden > 8 (C=256) always re-read."""
    from mga_yolo_amd.plan import PyramidPlan
    shapes = [(4, 64, 16, 16), (4, 128, 8, 8), (4, 256, 4, 4)]
    data, params, cfgs = [], [], []
    for l, (B, C, H, W) in enumerate(shapes):
        x, mask, gy = synth(B, C, H, W, seed=50 + l, mask_kind="mixed")
        p = O.Params.default_init(C, seed=l)
        p.beta.fill_(-0.4)
        data.append((x, mask, gy, p))
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta))
        cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
    plan = PyramidPlan(shapes, params, cfgs, want_gmask=want_gmask, use_proj=use_proj)
    for l, (x, mask, gy, _) in enumerate(data):
        plan.x[l].copy_(x); plan.mask[l].copy_(mask); plan.gy[l].copy_(gy)
    g = plan.capture(lambda: (plan.forward(), plan.backward()))
    g.replay(); g.replay()
    torch.cuda.synchronize()
    for l, (x, mask, gy, p) in enumerate(data):
        y_o, c = O.forward(x, mask, p)
        g_o = O.backward(gy, x, mask, p, O.Config(), c)
        assert rel_err(plan.y[l], y_o) < TOL and rel_err(plan.gx[l], g_o["gx"]) < TOL, l
        if want_gmask:
            assert rel_err(plan.gmask[l], g_o["gmask"]) < TOL, l
        for k, v in plan.named_param_grads(l).items():
            assert rel_err(v, g_o[k]) < TOL, (l, k)
    # split form (parameter gradients first, then input gradients) gives the same bits as the fused backward
    ref = [t.clone() for t in plan.gx] + [plan.grad_bucket.clone()]
    plan.forward(); plan.backward_params(); plan.backward_inputs()
    torch.cuda.synchronize()
    for a, b in zip(ref, list(plan.gx) + [plan.grad_bucket]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shapes,with_mask,fused", [
    ([(4, 64, 16, 16), (4, 128, 8, 8), (4, 256, 4, 4)], True, True),
    ([(32, 64, 80, 80), (32, 128, 40, 40), (32, 256, 20, 20)], True, True),  # BASELINE configs[1]: every CU busy, real waits
    ([(32, 128, 80, 80), (32, 256, 40, 40), (32, 512, 20, 20)], True, True), # BASELINE configs[2] per GPU: twice the workgroups, several resident rounds
    ([(3, 48, 17, 17), (5, 24, 9, 7)], True, True),                        # scalar (VEC=1) path, ragged, different B per level
    ([(9, 64, 40, 40), (9, 64, 20, 20)], False, True),                     # no mask; B not a multiple of the 8 XCDs
    ([(2, 256, 6, 160), (2, 64, 12, 80)], True, True),                     # a tile narrower than a row (64 px < W = 160): column-windowed staging
    ([(2, 256, 16, 160), (2, 512, 12, 80), (3, 512, 8, 40)], True, True),  # BASELINE configs[3] widths: 64- / 32-px tiles in 160- / 80- / 40-px rows (runs that wrap)
    ([(8, 256, 160, 160)], True, True),                                    # configs[3] P3 at full size: 3,200 narrow tiles, several resident rounds
    ([(1, 192, 9, 200), (2, 96, 5, 37)], False, True),                     # no mask; W not a multiple of the tile; scalar path
])
def test_fused_forward_launch_equals_three_launches(F, shapes, with_mask, fused, monkeypatch):
    """MGACBAM_FWD_FUSE: k_chan + k_apply as ONE x-resident launch (k_gate) with in-launch hand-off of the plane rows.
    Outputs and saved ctx fields must match the three-launch forward (same arithmetic per element; only the order of the
    channel-mean sum differs), the time-out word must stay clear and every tile flag must read the call count -- the
    next call depends on it -- also under graph replay."""
    from mga_yolo_amd import _lib
    from mga_yolo_amd.plan import PyramidPlan
    monkeypatch.setenv("MGACBAM_GATE_NARROW", "1")          # tiles narrower than an image row take k_gate too (opt-in: slower than the fallback)
    _lib.reload_env()
    try:
        _fused_vs_three(F, PyramidPlan, shapes, with_mask, fused)
    finally:
        monkeypatch.undo()
        _lib.reload_env()


def _fused_vs_three(F, PyramidPlan, shapes, with_mask, fused):
    params, cfgs = [], []
    for l, (B, C, H, W) in enumerate(shapes):
        p = O.Params.default_init(C, seed=l)
        p.beta.fill_(0.3)
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta))
        cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
    plans = [PyramidPlan(shapes, params, cfgs, with_mask=with_mask, want_gmask=False, fuse_forward=f) for f in (False, True)]
    gen = torch.Generator().manual_seed(7)
    for l, (B, C, H, W) in enumerate(shapes):
        x = torch.randn(B, C, H, W, generator=gen)
        m = torch.randn(B, 1, H, W, generator=gen) * 2
        m[0] = -40.0                                                        # sample 0: empty mask -> GAP fallback
        for pl in plans:
            pl.x[l].copy_(x)
            if with_mask:
                pl.mask[l].copy_(m)

    def check(calls):
        torch.cuda.synchronize()
        for l, (B, C, H, W) in enumerate(shapes):
            assert rel_err(plans[1].y[l], plans[0].y[l]) < 1e-6, (calls, l)
            a, b = plans[0].ctx_view(l), plans[1].ctx_view(l)
            for name in a:
                if name in ("sync", "proj"):
                    continue
                if a[name].dtype == torch.int32:
                    assert torch.equal(a[name], b[name]), (calls, l, name)
                else:
                    assert rel_err(b[name], a[name]) < 1e-6, (calls, l, name)
            sync = b["sync"]
            nf = B * ((H * W + 15) // 16 + 1)
            assert int(sync[nf:nf + 4].abs().sum()) == 0, (calls, l, "hand-off timed out")
            assert bool((sync[nf + 4:nf + 4 + B] == (calls if fused else 0)).all()), (calls, l, "per-sample ca flags")
            assert int(sync[nf + 4 + B:].abs().sum()) == 0, (calls, l, "no backward ran: its counters are untouched")
            flags = sync[:nf]
            want = calls if fused else 0                                    # ineligible groups never touch the flags
            assert int(flags.max()) == want and set(flags.unique().tolist()) <= {0, want}, (calls, l)

    plans[0].forward()
    for rep in range(3):
        plans[1].forward()
        check(rep + 1)
    g = plans[1].capture(plans[1].forward)          # capture warms up once (recording executes nothing)
    for l in range(len(shapes)):
        plans[1].y[l].zero_()
    g.replay(); g.replay()
    check(6)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)])
@pytest.mark.parametrize("shapes", [[(32, 64, 80, 80), (32, 128, 40, 40), (32, 256, 20, 20)],      # H*W % 8 == 0: 16-byte packed tiles
                                    [(3, 64, 6, 6), (3, 32, 10, 10)]])                              # H*W % 8 != 0: 8-byte tiles
def test_fused_forward_half_precision(F, shapes, dtype, tol):
    """k_gate with fp16 / bf16 features (kept packed in the registers, 8 elements per lane when H*W allows) against the
    three-launch forward and against the oracle on the same rounded inputs."""
    from mga_yolo_amd.plan import PyramidPlan
    params, cfgs, ps = [], [], []
    for l, (B, C, H, W) in enumerate(shapes):
        p = O.Params.default_init(C, seed=l)
        p.beta.fill_(0.2)
        ps.append(p)
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta))
        cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
    plans = [PyramidPlan(shapes, params, cfgs, dtype=dtype, want_gmask=False, fuse_forward=f) for f in (False, True)]
    gen = torch.Generator().manual_seed(17)
    data = []
    for l, s in enumerate(shapes):
        x = torch.randn(*s, generator=gen).to(dtype)
        m = torch.randn(s[0], 1, s[2], s[3], generator=gen)
        data.append((x, m))
        for pl in plans:
            pl.x[l].copy_(x); pl.mask[l].copy_(m)
    for pl in plans:
        pl.forward(); pl.forward()
    torch.cuda.synchronize()
    assert plans[1].gate_active()
    for l, (x, m) in enumerate(data):
        y_o, _ = O.forward(x.float(), m, ps[l])
        assert rel_err(plans[1].y[l].float(), y_o) < tol, l
        assert rel_err(plans[1].y[l].float(), plans[0].y[l].float()) < tol, l
        a, b = plans[0].ctx_view(l), plans[1].ctx_view(l)
        for name in ("planes", "sa", "ca"):
            assert rel_err(b[name], a[name]) < 1e-5, (l, name)
        assert torch.equal(a["cidx"], b["cidx"]), l


def test_fused_forward_generation_flags_wrap_around(F):
    """The hand-off flags count fused calls for the life of ctx; the compare is modulo 2^32, so crossing INT32_MAX is harmless."""
    from mga_yolo_amd.plan import PyramidPlan
    shapes = [(5, 64, 24, 24), (5, 128, 12, 12)]
    params, cfgs = [], []
    for l, (B, C, H, W) in enumerate(shapes):
        p = O.Params.default_init(C, seed=l)
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta))
        cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
    plans = [PyramidPlan(shapes, params, cfgs, want_gmask=False, fuse_forward=f) for f in (False, True)]
    gen = torch.Generator().manual_seed(3)
    for l, s in enumerate(shapes):
        x, m = torch.randn(*s, generator=gen), torch.randn(s[0], 1, s[2], s[3], generator=gen)
        for pl in plans:
            pl.x[l].copy_(x); pl.mask[l].copy_(m)
    plans[0].forward()
    for l, (B, C, H, W) in enumerate(shapes):
        sync = plans[1].ctx_view(l)["sync"]
        nf = B * ((H * W + 15) // 16 + 1)
        sync[:nf] = 0x7FFFFFFE                      # every flag two calls before the wrap (as after 2^31 - 2 fused calls)
        sync[nf + 4:nf + 4 + B] = 0x7FFFFFFE
    for rep in range(4):
        for l in range(len(shapes)):
            plans[1].y[l].zero_()
        plans[1].forward()
        torch.cuda.synchronize()
        for l, (B, C, H, W) in enumerate(shapes):
            assert rel_err(plans[1].y[l], plans[0].y[l]) < 1e-6, (rep, l)
            sync = plans[1].ctx_view(l)["sync"]
            nf = B * ((H * W + 15) // 16 + 1)
            assert int(sync[nf:nf + 4].abs().sum()) == 0, (rep, l, "hand-off timed out")
    want = (0x7FFFFFFE + 4) - (1 << 32)
    assert int(plans[1].ctx_view(0)["sync"][0]) == want


@pytest.mark.parametrize("k", [1, 9, 11, 15])
def test_generic_spatial_kernel_sizes(F, k):
    """spatial_k other than 3/5/7 takes the run-time-k code paths of the conv prologue, transposed conv and dWsa kernels."""
    B, C, H, W = 3, 32, 12, 20
    x, mask, gy = synth(B, C, H, W, seed=k, mask_kind="mixed")
    p = O.Params.default_init(C, k=k, seed=2)
    with torch.no_grad():
        p.wsa.mul_(3.0)
    y_o, c = O.forward(x, mask, p)
    g_o = O.backward(gy, x, mask, p, O.Config(), c)
    d = dict(x=x, mask=mask, gy=gy, params={"cam_mlp.0.weight": p.w1, "cam_mlp.0.bias": p.b1, "cam_mlp.2.weight": p.w2,
                                            "cam_mlp.2.bias": p.b2, "sam_conv.weight": p.wsa, "beta": p.beta},
             meta=dict(k=k, use_sigmoid_mask=True, tiny_thr=1e-4, eps=1e-6))
    y, g = _run_gpu(F, d)
    assert rel_err(y, y_o) < TOL
    for name in GRADS:
        assert rel_err(g[name], g_o[name]) < TOL, name


@pytest.mark.parametrize("shape,dtype,tol", [((2, 256, 160, 160), torch.bfloat16, 3e-2), ((2, 512, 40, 40), torch.float16, 4e-3),
                                             ((2, 512, 20, 20), torch.bfloat16, 3e-2), ((1, 768, 40, 40), torch.float32, 1e-4)])
def test_config5_shapes_low_precision(F, shape, dtype, tol):
    """BASELINE.json configs[4] (YOLOv8l widths, 640/1280 inputs, bf16) and the x-scale width: large C / hidden (32, 48) takes
    the general MLP paths; half-precision I/O with fp32 statistics, against the fp32 oracle on the rounded inputs."""
    B, C, H, W = shape
    x, mask, gy = synth(B, C, H, W, seed=91, mask_kind="sparse")
    x, gy = x.to(dtype).float(), gy.to(dtype).float()
    p = O.Params.default_init(C, seed=5)
    y_o, ctx = O.forward(x, mask, p)
    g_o = O.backward(gy, x, mask, p, O.Config(), ctx)
    xd = x.cuda().to(dtype).requires_grad_(True)
    md = mask.cuda().requires_grad_(True)
    ps = [t_.cuda().requires_grad_(True) for t_ in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
    y = F.mask_cbam(xd, md, *ps, F.BlockConfig(hidden=p.w1.shape[0]))
    y.backward(gy.cuda().to(dtype))
    assert rel_err(y.float(), y_o) < tol and rel_err(xd.grad.float(), g_o["gx"]) < tol and rel_err(md.grad, g_o["gmask"]) < tol
    for got, k in zip(ps, ("gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta")):
        assert rel_err(got.grad, g_o[k]) < tol, k


def test_reentrant_from_two_threads_on_two_streams(F):
    """The C ABI keeps no mutable global state: two host threads driving different shapes on different streams at once give
    the same bits as the same calls made one after the other."""
    import threading
    jobs = []
    for i, (B, C, H, W) in enumerate([(8, 64, 40, 40), (4, 128, 20, 20)]):
        x, mask, gy = synth(B, C, H, W, seed=60 + i)
        p = O.Params.default_init(C, seed=i)
        jobs.append(dict(x=x.cuda(), mask=mask.cuda(), gy=gy.cuda(), ps=[t_.cuda() for t_ in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)],
                         cfg=F.BlockConfig(hidden=p.w1.shape[0])))

    def run(j, stream, out):
        with torch.cuda.stream(stream):
            for _ in range(20):
                x = j["x"].clone().requires_grad_(True)
                m = j["mask"].clone().requires_grad_(True)
                y = F.mask_cbam(x, m, *j["ps"], j["cfg"])
                y.backward(j["gy"])
            out.append((y.detach().clone(), x.grad.clone(), m.grad.clone()))
        stream.synchronize()

    serial = [[], []]
    for j, o in zip(jobs, serial):
        run(j, torch.cuda.current_stream(), o)
    torch.cuda.synchronize()
    par = [[], []]
    ths = [threading.Thread(target=run, args=(j, torch.cuda.Stream(), o)) for j, o in zip(jobs, par)]
    for t_ in ths:
        t_.start()
    for t_ in ths:
        t_.join()
    torch.cuda.synchronize()
    for s_, p_ in zip(serial, par):
        for a, b in zip(s_[0], p_[0]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("B,C,H,W,k,r,kind", [
    (3, 64, 14, 27, 1, 16, "mixed"),      # k = 1, H*W odd: the conv prologue stages a 1-row window (divide-by-one index magic)
    (16, 130, 23, 1, 1, 16, "sparse"),    # W = 1: 1-column windows
    (9, 256, 28, 1, 7, 1, "randn"),       # r = 1: hidden = C = 256 with one lane per row (LDS chunking of the hidden partials)
    (2, 320, 2, 12, 3, 1, "randn"),
    (8, 320, 8, 3, 9, 1, "prob"),
    (5, 3, 23, 17, 1, 1, "mixed"),
])
def test_regressions_found_by_the_fuzzer(F, B, C, H, W, k, r, kind):
    """Shapes that tests/fuzz/fuzz_parity.py (randomised parity fuzz, 1000 cases on MI355X) once broke."""
    x, mask, gy = synth(B, C, H, W, seed=5, mask_kind=kind)
    p = O.Params.default_init(C, r=r, k=k, seed=1)
    cfg = O.Config(use_sigmoid_mask=kind != "prob")
    y_o, c = O.forward(x, mask, p, cfg)
    g_o = O.backward(gy, x, mask, p, cfg, c)
    d = dict(x=x, mask=mask, gy=gy, params={"cam_mlp.0.weight": p.w1, "cam_mlp.0.bias": p.b1, "cam_mlp.2.weight": p.w2,
                                            "cam_mlp.2.bias": p.b2, "sam_conv.weight": p.wsa, "beta": p.beta},
             meta=dict(k=k, use_sigmoid_mask=kind != "prob", tiny_thr=1e-4, eps=1e-6))
    y, g = _run_gpu(F, d)
    assert rel_err(y, y_o) < TOL
    floor = 1e-6 * float(gy.norm() * x.norm())
    for name in GRADS:
        tol = TOL * float(g_o[name].abs().max()) + (floor if name not in ("gx",) else 0.0)
        assert float((g[name].detach().cpu().double() - g_o[name].double()).abs().max()) <= tol, name


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1..4] as workloads: every config runs as ONE pyramid call (P3+P4+P5 in one library call each way) at its
# per-GPU batch, against checksums the reference itself produced on the same seeded inputs (oracle/gen_golden.py --big-only)
# ---------------------------------------------------------------------------------------------------------------------------
CONFIG_LEVELS = {
    "cfg2": ("YOLOv8n, 32 x 640^2 per GPU, fp32", ["cfg2_p3", "cfg2_p4", "cfg2_p5"], torch.float32),
    "cfg3": ("YOLOv8s, batch 256 over 8 GPUs = 32 per GPU, fp32", ["cfg3_p3", "cfg3_p4", "cfg3_p5"], torch.float32),
    "cfg4": ("YOLOv8m @1280, 8 per GPU, fp32", ["cfg4_p3", "cfg4_p4", "cfg4_p5"], torch.float32),
    "cfg5_640": ("YOLOv8l @640, bf16", ["cfg5_640_p3", "cfg5_640_p4", "cfg5_640_p5"], torch.bfloat16),
    "cfg5_1280": ("YOLOv8l @1280, bf16", ["cfg5_1280_p3", "cfg5_1280_p4", "cfg5_1280_p5"], torch.bfloat16),
}


@pytest.mark.parametrize("config", list(CONFIG_LEVELS))
def test_baseline_config_as_one_pyramid_call_vs_reference_checksums(F, checksums, config):
    _, names, dtype = CONFIG_LEVELS[config]
    tol = TOL if dtype == torch.float32 else 1e-3        # bf16: element rounding 2^-9, unbiased -> sums agree far better than elements
    levels, leaves, gys = [], [], []
    for name in names:
        ref = checksums["big"][name]
        B, C, H, W = ref["shape"]
        x, mask, gy = synth(B, C, H, W, mask_kind=ref["mask_kind"])
        p = O.Params.default_init(C)
        ps = [t.cuda().requires_grad_(True) for t in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
        xd = x.cuda().to(dtype).requires_grad_(True)
        md = mask.cuda().requires_grad_(True)
        levels.append((xd, md, ps, F.BlockConfig(hidden=p.w1.shape[0])))
        leaves.append((xd, md, ps))
        gys.append(gy.cuda().to(dtype))
        del x, mask, gy

    def run():
        for xd, md, ps in leaves:
            xd.grad = md.grad = None
            for t in ps:
                t.grad = None
        ys = F.mask_cbam_pyramid(levels)                  # ONE library call forward ...
        torch.autograd.backward(ys, gys)                  # ... and one backward for the three levels
        torch.cuda.synchronize()
        return [dict(y=y.detach(), gx=xd.grad, gmask=md.grad, gw1=ps[0].grad, gb1=ps[1].grad, gw2=ps[2].grad, gb2=ps[3].grad,
                     gwsa=ps[4].grad, gbeta=ps[5].grad) for y, (xd, md, ps) in zip(ys, leaves)]

    got = run()
    report = []
    for name, g in zip(names, got):
        ref = checksums["big"][name]
        for k, v in g.items():
            c = checksum(v.float())
            scale = ref[k]["abs"] + 1e-12
            for f in ("sum", "wsum", "abs"):
                if not abs(c[f] - ref[k][f]) <= tol * scale:
                    report.append(f"{name}.{k}.{f}: got {c[f]:.6f} want {ref[k][f]:.6f}")
    assert not report, f"{config}: " + "; ".join(report)
    # size-independent properties at full size: bitwise run-to-run reproducibility, linearity in the upstream gradient
    snap = [{k: v.clone() for k, v in g.items()} for g in got]
    again = run()
    for a, b in zip(snap, again):
        assert all(torch.equal(a[k], b[k]) for k in a)
    if dtype == torch.float32:
        gys[:] = [-2.5 * g for g in gys]
        scaled = run()
        for a, b in zip(snap, scaled):
            for k in GRADS:
                assert rel_err(b[k], -2.5 * a[k]) < 1e-5, (config, k)
    from mga_yolo_amd import handoff_report
    assert handoff_report() >= 3                          # every in-launch hand-off of every call above completed


# ---------------------------------------------------------------------------------------------------------------------------
# the in-launch hand-off must fail loudly and be sized from the device
# ---------------------------------------------------------------------------------------------------------------------------
def _small_plan(F, fuse=True, shapes=((4, 64, 16, 16), (4, 128, 8, 8))):
    from mga_yolo_amd.plan import PyramidPlan
    params, cfgs = [], []
    for l, (B, C, H, W) in enumerate(shapes):
        p = O.Params.default_init(C, seed=l)
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta))
        cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
    plan = PyramidPlan(list(shapes), params, cfgs, fuse_forward=fuse)
    gen = torch.Generator().manual_seed(11)
    for l, s_ in enumerate(shapes):
        plan.x[l].copy_(torch.randn(*s_, generator=gen))
        plan.mask[l].copy_(torch.randn(s_[0], 1, s_[2], s_[3], generator=gen))
        plan.gy[l].copy_(torch.randn(*s_, generator=gen))
    return plan


def test_handoff_eligibility_is_sized_from_the_device(F, monkeypatch):
    """A residency budget too small for the tiles a wait spans (here forced through the override of CUs x occupancy) makes the
    library choose the three-launch forward; results are the same."""
    from mga_yolo_amd import _lib
    plan = _small_plan(F)
    plan.forward()
    assert plan.gate_active()
    y_gate = [t.clone() for t in plan.y]
    monkeypatch.setenv("MGACBAM_RESIDENT_WGS", "8")       # as on a device / partition with room for 8 workgroups
    _lib.reload_env()
    try:
        small = _small_plan(F)
        small.forward()
        assert not small.gate_active()                    # k_chan + k_apply ran: the hand-off flags were never touched
        small.check_handoff()
        for a, b in zip(small.y, y_gate):
            assert rel_err(a, b) < 1e-6
    finally:
        monkeypatch.undo()
        _lib.reload_env()
    again = _small_plan(F)
    again.forward()
    assert again.gate_active()


def test_handoff_timeout_is_loud(F, monkeypatch):
    """Fault injection: sample 0's role workgroup never publishes ca.  Its tiles time out (bounded spin), the launch still drains,
    the status word is set, the product path raises, and the affected outputs are NaN -- not silently stale."""
    from mga_yolo_amd import _lib, HandoffTimeout
    monkeypatch.setenv("MGACBAM_FAULT", "1")
    monkeypatch.setenv("MGACBAM_SPIN_LIMIT", "2000")
    _lib.reload_env()
    try:
        plan = _small_plan(F)
        plan.forward()
        with pytest.raises(HandoffTimeout):
            plan.check_handoff()
        for l in range(plan.n):
            assert bool(torch.isnan(plan.y[l][0]).all()), l               # sample 0: poisoned
            assert not bool(torch.isnan(plan.y[l][1:]).any()), l          # the other samples completed normally
    finally:
        monkeypatch.undo()
        _lib.reload_env()
    good = _small_plan(F)
    good.forward(); good.backward()
    good.check_handoff()
    assert not any(bool(torch.isnan(t).any()) for t in good.y + good.gx)


def test_fold_counters_survive_a_half_finished_backward(F):
    """MGACBAM_BWD_FOLD's flags are generation counters that nothing resets: a backward that stops after the folded launch (error,
    split-stage caller) leaves a consistent state and the next full step is still right."""
    from mga_yolo_amd import _lib
    plan = _small_plan(F, shapes=((4, 64, 40, 40), (4, 128, 20, 20)))
    ref = _small_plan(F, shapes=((4, 64, 40, 40), (4, 128, 20, 20)))
    assert plan.fold_active()
    B = _lib.BWD_STAGES
    plan.forward()
    for _ in range(3):                                    # three folded launches in a row, never followed by the apply stage
        plan.backward(B["reduce1"] | B["convT"] | _lib.BWD_FOLD)
    plan.forward(); plan.backward()
    ref.forward(); ref.backward()
    plan.check_handoff(); ref.check_handoff()
    for a, b in zip(plan.gx + plan.gmask + [plan.grad_bucket], ref.gx + ref.gmask + [ref.grad_bucket]):
        assert torch.equal(a, b)
    for l, (Bn, C, H, W) in enumerate(plan.shapes):
        nf = Bn * ((H * W + 15) // 16 + 1)
        sync = plan.ctx_view(l)["sync"]
        o = nf + 4 + Bn
        tiles, convs = sync[o:o + nf], sync[o + nf:o + 2 * nf]
        assert set(tiles.unique().tolist()) <= {0, 4} and int(tiles.max()) == 4      # fold_active's launch + 3 folded launches
        assert set(convs.unique().tolist()) <= {0, 4} and int(convs.max()) == 4
        # the full backward ran as the MERGED launch (k_bwd_r12), which counts in generation counters of its own -- one class per kind
        # of workgroup, each bumped exactly once per merged launch -- so the two launch forms can alternate on one ctx
        for name, blk in (("tiles", sync[o + 2 * nf:o + 3 * nf]), ("conv tiles", sync[o + 3 * nf:o + 4 * nf]),
                          ("dWsa tiles", sync[o + 4 * nf:o + 5 * nf]), ("sweeps", sync[o + 5 * nf:])):
            assert set(blk.unique().tolist()) <= {0, 1} and int(blk.max()) == 1, (l, name)
    plan.forward(); plan.backward()                       # and once more: merged generation 2 next to fold generation 4
    ref.forward(); ref.backward()
    plan.check_handoff()
    for a, b in zip(plan.gx + plan.gmask + [plan.grad_bucket], ref.gx + ref.gmask + [ref.grad_bucket]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shapes", [[(32, 64, 80, 80), (32, 128, 40, 40), (32, 256, 20, 20)], [(5, 48, 24, 40), (3, 96, 12, 20)]])
def test_merged_backward_launch_gives_the_same_bits(F, shapes, monkeypatch):
    """k_bwd_r12 (k_bwd_reduce1 tiles + transposed-conv tiles + dWsa tiles + k_bwd_reduce2 sweeps in ONE launch, in-launch hand-offs per
    sample) against the separate launches (MGACBAM_BWD_MERGE=0): every gradient bit for bit, eager calls and graph replays, status clear."""
    from mga_yolo_amd import _lib
    from mga_yolo_amd.plan import PyramidPlan
    params, cfgs, data = [], [], []
    for l, (B, C, H, W) in enumerate(shapes):
        p = O.Params.default_init(C, seed=l)
        p.beta.fill_(0.25)
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)); cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
        data.append(synth(B, C, H, W, seed=80 + l, mask_kind="sparse"))

    def run():
        plan = PyramidPlan(shapes, params, cfgs)
        for l, (x, mask, gy) in enumerate(data):
            plan.x[l].copy_(x); plan.mask[l].copy_(mask); plan.gy[l].copy_(gy)
        for _ in range(2):
            plan.forward(); plan.backward()
        g = plan.capture(lambda: (plan.forward(), plan.backward()))
        for _ in range(3):
            g.replay()
        plan.check_handoff()
        return plan.grad_bucket.clone(), [t.clone() for t in plan.gx], [t.clone() for t in plan.gmask]

    merged = run()
    monkeypatch.setenv("MGACBAM_BWD_MERGE", "0")
    _lib.reload_env()
    try:
        split = run()
    finally:
        monkeypatch.undo()
        _lib.reload_env()
    assert torch.equal(merged[0], split[0])
    assert all(torch.equal(a, b) for a, b in zip(merged[1] + merged[2], split[1] + split[2]))


def test_stale_scratch_size_after_a_knob_change_raises_instead_of_faulting(F, monkeypatch):
    """Round 2's GPU abort (rc 134): a scratch size cached across mgacbam_reload_env() was too small for the new launch geometry and the
    kernels wrote out of bounds.  ABI 14: the plan's buffers travel with their capacities, the library recomputes the requirement under
    the current knobs and refuses (MGACBAM_E_SIZE) -- nothing is launched; with the knobs restored the same plan runs and is correct."""
    from mga_yolo_amd import _lib
    from mga_yolo_amd.plan import PyramidPlan
    shapes = [(4, 64, 40, 40)]
    x, mask, gy = synth(4, 64, 40, 40, seed=31)
    p = O.Params.default_init(64, seed=2)
    plan = PyramidPlan(shapes, [(p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)], [F.BlockConfig(hidden=p.w1.shape[0])])
    plan.x[0].copy_(x); plan.mask[0].copy_(mask); plan.gy[0].copy_(gy)
    plan.forward(); plan.backward()
    torch.cuda.synchronize()
    ref_gx = plan.gx[0].clone()
    monkeypatch.setenv("MGACBAM_CHAN_TX", "1")                    # one pixel vector per tile: ~16x the tile partials
    _lib.load().mgacbam_reload_env()                              # the raw entry point: the plan keeps its (now stale) buffer sizes
    try:
        plan.forward()
        with pytest.raises(RuntimeError, match="too small"):
            plan.backward()
        torch.cuda.synchronize()
    finally:
        monkeypatch.undo()
        _lib.reload_env()
    plan.forward(); plan.backward()
    torch.cuda.synchronize()
    assert torch.equal(plan.gx[0], ref_gx)
    y_o, c = O.forward(x, mask, p)
    assert rel_err(plan.gx[0], O.backward(gy, x, mask, p, O.Config(), c)["gx"]) < TOL


@pytest.mark.parametrize("mdtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_half_precision_mask_gets_its_gradient_in_its_own_dtype(F, mdtype, tol):
    """Under AMP the mask logits arrive in half precision (the mask head's conv runs under autocast): dL/dmask must come back in that
    dtype with the right VALUES.  Regression: the cast of dL/dmask to the mask's dtype was enqueued before the launch that writes it
    (it read unwritten memory; invisible with fp32 masks, where the cast is the identity) -- found by the AMP slice test of round 3."""
    from mga_yolo_amd.functional import EcaConfig, mask_eca
    B, C, H, W = 4, 64, 20, 20
    x, mask, gy = synth(B, C, H, W, seed=23)
    mask = mask.to(mdtype).float()
    p = O.Params.default_init(C)
    y_o, ctx = O.forward(x, mask, p)
    g_o = O.backward(gy, x, mask, p, O.Config(), ctx)
    for rep in range(3):                                           # (fresh allocations each time: stale contents differ)
        xd = x.cuda().requires_grad_(True)
        md = mask.cuda().to(mdtype).requires_grad_(True)
        ps = [t.cuda() for t in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
        y = F.mask_cbam(xd, md, *ps, F.BlockConfig(hidden=p.w1.shape[0]))
        torch.cuda.empty_cache()
        y.backward(gy.cuda())
        assert md.grad.dtype == mdtype and md.grad.shape == md.shape
        assert rel_err(md.grad.float(), g_o["gmask"]) < tol and rel_err(xd.grad, g_o["gx"]) < TOL
    from oracle import maskeca_oracle as EO
    we = torch.randn(1, 1, 3, generator=torch.Generator().manual_seed(4))
    be = torch.tensor(0.2)
    xd = x.cuda().requires_grad_(True)
    md = mask.cuda().to(mdtype).requires_grad_(True)
    ye = mask_eca(xd, md, we.cuda(), be.cuda(), EcaConfig(k=3))
    ye.backward(gy.cuda())
    xh, mh = x.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    from mga_yolo_amd.module import _eca_host_forward
    _eca_host_forward(xh, mh, we, be, EcaConfig(k=3)).backward(gy)
    assert md.grad.dtype == mdtype and rel_err(md.grad.float(), mh.grad) < tol and rel_err(xd.grad, xh.grad) < TOL


def test_dwsa_tail_roles_give_the_same_bits(F, monkeypatch):
    """MGACBAM_WSA_TAIL=1 (opt-in): the dWsa tile partials and their fixed-order sums ride at the END of the k_bwd_apply launch behind an
    arrival counter instead of at the front of k_bwd_reduce2 -- every gradient must equal the default placement bit for bit, over
    repeated calls and graph replays (the counters return to 0 each call), and the hand-off status must stay clear."""
    from mga_yolo_amd import _lib
    from mga_yolo_amd.plan import PyramidPlan
    shapes = [(8, 64, 40, 40), (8, 128, 20, 20), (5, 256, 10, 10)]
    params, cfgs, data = [], [], []
    for l, (B, C, H, W) in enumerate(shapes):
        p = O.Params.default_init(C, seed=l)
        p.beta.fill_(0.2)
        params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta))
        cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
        data.append(synth(B, C, H, W, seed=60 + l, mask_kind="mixed"))

    def run():
        plan = PyramidPlan(shapes, params, cfgs)
        for l, (x, mask, gy) in enumerate(data):
            plan.x[l].copy_(x); plan.mask[l].copy_(mask); plan.gy[l].copy_(gy)
        for _ in range(3):
            plan.forward(); plan.backward()
        g = plan.capture(lambda: (plan.forward(), plan.backward()))
        g.replay(); g.replay()
        plan.check_handoff()
        words = [plan.ctx_view(l)["sync"][B * ((H * W + 15) // 16 + 1):][:4].clone() for l, (B, C, H, W) in enumerate(shapes)]
        return plan.grad_bucket.clone(), [t.clone() for t in plan.gx], words

    base = run()
    monkeypatch.setenv("MGACBAM_WSA_TAIL", "1")
    _lib.reload_env()
    try:
        tail = run()
    finally:
        monkeypatch.undo()
        _lib.reload_env()
    assert torch.equal(base[0], tail[0]) and all(torch.equal(a, b) for a, b in zip(base[1], tail[1]))
    assert all(int(w.abs().sum()) == 0 for w in tail[2]), "arrival counters must be back at 0 between calls"


def test_rccl_branch_of_the_gradient_exchange_runs_on_this_gpu(F):
    """`GradExchange`'s "nccl" (= RCCL) branch -- ReduceOp.AVG inside the collective, launched on a side stream behind an event, joined
    back into the compute stream -- on hardware.  A one-GPU box cannot host two RCCL ranks, so this is a ONE-rank group with the
    collective forced (numerically the identity): it proves the branch executes and orders correctly against the compute stream (the
    bucket is written by a kernel right before start() and read right after finish()), not that two GPUs agree -- that needs the node."""
    import os
    import torch.distributed as dist
    from mga_yolo_amd.dp import GradExchange, payload_buckets
    from mga_yolo_amd.plan import PyramidPlan
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29650 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        shapes = [(4, 64, 20, 20), (4, 128, 10, 10)]
        params, cfgs = [], []
        for l, (B, C, H, W) in enumerate(shapes):
            p = O.Params.default_init(C, seed=l)
            params.append((p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)); cfgs.append(F.BlockConfig(hidden=p.w1.shape[0]))
        plan = PyramidPlan(shapes, params, cfgs)
        for l, (B, C, H, W) in enumerate(shapes):
            x, mask, gy = synth(B, C, H, W, seed=70 + l)
            plan.x[l].copy_(x); plan.mask[l].copy_(mask); plan.gy[l].copy_(gy)
        plan.forward(); plan.backward()
        torch.cuda.synchronize()
        want = plan.grad_bucket.clone()
        payload = payload_buckets(3 * (1 << 20), plan.grad_bucket.device, cap_mb=1.0)      # three 1 MB buckets beside the blocks' own
        for i, b in enumerate(payload):
            b.fill_(float(i + 1))
        ex = GradExchange([plan.grad_bucket] + payload, single_rank_collective=True)
        assert ex.avg_in_collective and ex.side is not None
        for _ in range(3):
            plan.grad_bucket.zero_()
            plan.forward(); plan.backward()                    # writes the bucket on the compute stream ...
            ex.start()                                         # ... the collective waits for exactly that point
            plan.forward(_lib_pool())                          # parameter-free work overlapping the exchange, as bench.py does
            ex.finish()
            got = plan.grad_bucket.clone()                     # ordered after the collective by finish()
            torch.cuda.synchronize()
            assert torch.equal(got, want)
        assert all(float(b[0]) == i + 1 and float(b[-1]) == i + 1 for i, b in enumerate(payload))
    finally:
        dist.destroy_process_group()


def _lib_pool():
    from mga_yolo_amd import _lib
    return _lib.FWD_STAGES["pool"]
