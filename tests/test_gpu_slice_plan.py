"""SlicePlan (mga_yolo_amd/slice.py): the layer-loop slice of a training step -- mask heads -> MaskCBAM -> segmentation loss -> Kendall
combine and the backward of all of it -- as graph-captured C-ABI calls on static buffers, against the same slice composed from this
package's modules through autograd (which the other GPU tests pin to the reference's goldens)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _build(shapes, hidden, seed=0):
    from mga_yolo_amd import MGAMaskHead, MaskCBAM
    heads, blocks = [], []
    for l, ((B, C, H, W), hid) in enumerate(zip(shapes, hidden)):
        torch.manual_seed(seed + l)
        h = MGAMaskHead(C, hid)
        h.proj[1].eps, h.proj[1].momentum = 1e-3, 0.03
        torch.manual_seed(seed + 10 + l)
        b = MaskCBAM(C)
        with torch.no_grad():
            b.beta.fill_(0.2 * (l - 1))
        heads.append(h.cuda().train()); blocks.append(b.cuda())
    return heads, blocks


@pytest.mark.parametrize("shapes,hidden,target_hw", [
    ([(4, 64, 16, 16), (4, 128, 8, 8), (4, 256, 4, 4)], [16, 32, 64], None),
    ([(3, 64, 20, 12), (3, 128, 10, 6)], [16, 32], [(80, 48), (80, 48)]),            # nearest-resized full-resolution targets, 2 levels
])
def test_slice_plan_equals_the_module_composition(built_lib, shapes, hidden, target_hw):
    from mga_yolo_amd import SegLossConfig, SegmentationLoss, kendall_combine
    from mga_yolo_amd.slice import HEAD_PARAM_NAMES, SlicePlan
    heads, blocks = _build(shapes, hidden)
    plan = SlicePlan(shapes, hidden, [b.block_params() for b in blocks], [b.block_config() for b in blocks],
                     [{k: v.detach().clone() for k, v in h.state_dict().items()} for h in heads], target_hw=target_hw,
                     scale_weights=(1.0, 0.5, 2.0))
    g = torch.Generator().manual_seed(21)
    xs, gys, tgs = [], [], []
    for l, (B, C, H, W) in enumerate(shapes):
        th, tw = (H, W) if target_hw is None else target_hw[l]
        xs.append(torch.randn(B, C, H, W, generator=g).cuda())
        gys.append(torch.randn(B, C, H, W, generator=g).cuda())
        tgs.append((torch.rand(B, 1, th, tw, generator=g) > 0.7).float().cuda())
        plan.x[l].copy_(xs[l]); plan.gy[l].copy_(gys[l]); plan.targets[l].copy_(tgs[l])
    det = torch.tensor([1.3, 0.7, 2.1]).cuda()
    lv = torch.tensor([0.3, -0.4]).cuda()
    plan.det_loss.copy_(det); plan.log_vars.copy_(lv)
    graph = plan.capture(plan.step)
    for rm, rv, nbt in plan.head_buffers:                          # capture's warm-up run was a training step too: start over
        rm.zero_(); rv.fill_(1.0); nbt.zero_()
    graph.replay()
    torch.cuda.synchronize()
    plan.check_handoff()
    # ---- the same slice through the modules + autograd --------------------------------------------------------------------------------
    for h in heads:
        h.proj[1].reset_running_stats()
    xl = [x.clone().requires_grad_(True) for x in xs]
    lvl = lv.clone().requires_grad_(True)
    logits = [h(x) for h, x in zip(heads, xl)]
    ys = [b([x, m]) for b, x, m in zip(blocks, xl, logits)]
    crit = SegmentationLoss(SegLossConfig(scale_weights=(1.0, 0.5, 2.0)))
    seg_total, logs = crit({k: m for k, m in zip(("p3", "p4", "p5"), logits)}, tgs)
    total = kendall_combine(det, seg_total, lvl)
    torch.autograd.backward([total.sum()] + ys, [None] + gys)
    torch.cuda.synchronize()
    assert rel_err(plan.total, total) < 1e-6 and abs(float(plan.seg_out[0]) - logs["seg_total"]) < 1e-5
    assert rel_err(plan.g_log_vars, lvl.grad) < 1e-5
    for l in range(len(shapes)):
        assert rel_err(plan.logits[l], logits[l]) < 1e-6 and rel_err(plan.y[l], ys[l]) < 1e-6, l
        assert rel_err(plan.gx[l], xl[l].grad) < 1e-5, l            # MaskCBAM's part + the head's part, accumulated in the GEMM epilogue
        sd = dict(heads[l].named_parameters())
        for k, gq in zip(HEAD_PARAM_NAMES, plan.head_grads[l]):
            assert rel_err(gq, sd[k].grad) < 1e-5, (l, k)
        for (name, gq), p in zip(plan.cbam.named_param_grads(l).items(), blocks[l].block_params()):
            assert rel_err(gq, p.grad) < 1e-5, (l, name)
        assert rel_err(plan.head_buffers[l][0], heads[l].proj[1].running_mean) < 1e-6
        assert rel_err(plan.head_buffers[l][1], heads[l].proj[1].running_var) < 1e-6
    # replaying the graph again is a new training step on the same inputs: same outputs, running statistics move on
    rm = plan.head_buffers[0][0].clone()
    y0 = plan.y[0].clone()
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(plan.y[0], y0) and not torch.equal(plan.head_buffers[0][0], rm)
    assert int(plan.head_buffers[0][2]) == 2


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.bfloat16, 3e-2)])
def test_slice_plan_with_half_precision_features(built_lib, dtype, tol):
    """SlicePlan(dtype=fp16 | bf16): half-precision features and feature gradients, fp32 mask logits straight from the heads
    (MGAHEAD_LOGITS_F32: no conversion pass between head, block and loss), fp32 statistics and parameter gradients -- against the fp32
    plan on the same rounded inputs, at the tolerances of the half-precision I/O tests (BASELINE configs[4] is a bf16 config)."""
    from mga_yolo_amd.slice import SlicePlan
    shapes, hidden = [(4, 64, 16, 16), (4, 128, 8, 8), (4, 256, 4, 4)], [16, 32, 64]
    heads, blocks = _build(shapes, hidden)
    mk = lambda dt: SlicePlan(shapes, hidden, [b.block_params() for b in blocks], [b.block_config() for b in blocks],
                              [{k: v.detach().clone() for k, v in h.state_dict().items()} for h in heads], scale_weights=(1.0, 0.5, 2.0), dtype=dt)
    ref, plan = mk(torch.float32), mk(dtype)
    assert plan.x[0].dtype == dtype and plan.gx[0].dtype == dtype and plan.logits[0].dtype == torch.float32
    g = torch.Generator().manual_seed(33)
    for l, (B, C, H, W) in enumerate(shapes):
        x = torch.randn(B, C, H, W, generator=g).to(dtype)
        gy = torch.randn(B, C, H, W, generator=g).to(dtype)
        t = (torch.rand(B, 1, H, W, generator=g) > 0.7).float()
        for p_ in (ref, plan):
            p_.x[l].copy_(x); p_.gy[l].copy_(gy); p_.targets[l].copy_(t)
    for p_ in (ref, plan):
        p_.det_loss.copy_(torch.tensor([1.3, 0.7, 2.1])); p_.log_vars.copy_(torch.tensor([0.3, -0.4]))
        p_.step()
    torch.cuda.synchronize()
    plan.check_handoff()
    assert rel_err(plan.total, ref.total) < tol
    for l in range(len(shapes)):
        assert rel_err(plan.logits[l], ref.logits[l]) < tol and rel_err(plan.y[l].float(), ref.y[l]) < tol, l
        assert rel_err(plan.gx[l].float(), ref.gx[l]) < 2 * tol, l
    assert rel_err(plan.grad_bucket, ref.grad_bucket) < 2 * tol
    g2 = plan.capture(plan.step)                                       # graph-capturable like the fp32 plan
    g2.replay(); torch.cuda.synchronize()
    plan.check_handoff()
