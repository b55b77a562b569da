"""The drop-in proof against the reference's REAL model factory, recorded by ``oracle/check_dropin.py`` in the build container
(the reference never travels): ``tests/golden/dropin_report.json``.  This test asserts on the recorded outcome and re-checks,
without the reference, what the recorded keys / shapes allow: that this package's classes really have that state_dict."""
import json
import os

import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def report():
    with open(os.path.join(GOLDEN, "dropin_report.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("scale", ["n", "s"])
def test_reference_factory_built_our_blocks(report, scale):
    R = report["scales"][scale]
    # install() ran AFTER the reference imports and reached the namespace whose parse_model builds MGAModel
    assert "mga_yolo.external.ultralytics.ultralytics.nn.tasks" in R["patched_modules"]
    assert "ultralytics.nn.tasks" in R["patched_modules"] and "mga_yolo.nn.modules.masked_cbam" in R["patched_modules"]
    assert R["blocks_are_ours"] and R["late_import_is_ours"] and R["deepcopy_ok"]
    for i, feat, head in ((23, 15, 22), (25, 18, 24), (27, 21, 26)):
        L = R["layers"][str(i)]
        assert L["cls"] == "mga_yolo_amd.module.MaskCBAM" and L["i"] == i and L["f"] == [feat, head]      # U/nn/tasks.py:1763-1766
        assert L["type"].endswith("MaskCBAM") and L["np"] == L["ref_np"]
    assert R["detect_from"] == [23, 25, 27]
    # the mask heads (layers 22/24/26, U/nn/tasks.py:1724-1731) are this package's MGAMaskHead too, and MGAModel._index_mask_heads
    # (isinstance against the patched name, model/model.py:208-214) still finds them: the `seg` outputs were compared above
    assert R["mask_heads"] == ["mga_yolo_amd.module.MGAMaskHead"] * 3
    for i, feat in ((22, 15), (24, 18), (26, 21)):
        L = R["layers"][str(i)]
        assert L["cls"] == "mga_yolo_amd.module.MGAMaskHead" and L["f"] == feat and L["np"] == L["ref_np"]
    assert R["state_keys_equal"] and R["state_values_equal_same_seed"] and R["cross_load_strict"]
    assert R["eval_forward_rel_diff"] <= 1e-6 and R["train_forward_rel_diff"] <= 1e-6
    assert R["seg_criterion"]["ref"].startswith("mga_yolo.nn.losses") and R["seg_criterion"]["new"] == "mga_yolo_amd.segloss.SegmentationLoss"
    assert abs(R["loss_new"] - R["loss_ref"]) <= 1e-5 * abs(R["loss_ref"])
    assert R["grad_keys_equal"] and R["grad_worst_rel_diff"] < 1e-4
    assert all(v < 1e-4 for v in R["block_grad_rel_diff"].values()) and len(R["block_grad_rel_diff"]) == 18
    assert len(R["alpha_found"]) == 3 and all(0.5 < a < 1.0 for a in R["alpha_found"])


@pytest.mark.parametrize("scale", ["n", "s"])
def test_our_classes_have_the_recorded_state(report, scale):
    from mga_yolo_amd import MaskCBAM
    R = report["scales"][scale]
    for i in (23, 25, 27):
        L = R["layers"][str(i)]
        C = L["state"]["cam_mlp.2.bias"][0]
        m = MaskCBAM(C)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == L["state"]
        assert sum(p.numel() for p in m.parameters()) == L["np"]
    heads = {22: R["layers"]["22"], 24: R["layers"]["24"], 26: R["layers"]["26"]}
    for i, L in heads.items():                                   # MGAMaskHead: proj.0 / proj.1 / head
        assert set(L["state"]) == {"proj.0.weight", "proj.1.weight", "proj.1.bias", "proj.1.running_mean", "proj.1.running_var",
                                   "proj.1.num_batches_tracked", "head.weight", "head.bias"}
        try:
            from mga_yolo_amd import MGAMaskHead
        except ImportError:
            continue
        hid, cin = L["state"]["proj.0.weight"][:2]
        h = MGAMaskHead(cin, hid)
        assert {k: list(v.shape) for k, v in h.state_dict().items()} == L["state"]
        assert sum(p.numel() for p in h.parameters()) == L["np"]
