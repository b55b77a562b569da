"""Pins oracle/maskcbam_oracle.py to outputs of the reference itself (tests/golden, made by oracle/gen_golden.py)."""
import pytest
import torch

from conftest import checksum, golden_case_names, load_golden, rel_err, synth
from oracle import maskcbam_oracle as O

GRADS = ("gx", "gmask", "gw1", "gb1", "gw2", "gb2", "gwsa", "gbeta")
TOL = 2e-5   # fp32, same math in a different summation order


def _run_explicit(d, dtype=torch.float32):
    p = O.Params.from_state_dict(d["params"]).to(dtype)
    cfg = O.Config(use_sigmoid_mask=d["meta"]["use_sigmoid_mask"], tiny_thr=d["meta"]["tiny_thr"], eps=d["meta"]["eps"])
    x = d["x"].to(dtype)
    mask = None if d["mask"] is None else d["mask"].to(dtype)
    y, ctx = O.forward(x, mask, p, cfg)
    g = O.backward(d["gy"].to(dtype), x, mask, p, cfg, ctx)
    return y, g, ctx


@pytest.mark.parametrize("name", golden_case_names())
def test_explicit_matches_reference(name):
    d = load_golden(name)
    y, g, _ = _run_explicit(d)
    assert rel_err(y, d["out"]["y"]) < TOL
    for k in GRADS:
        if k == "gmask" and d["mask"] is None:
            assert g[k] is None
            continue
        assert g[k].shape == d["out"][k].shape, k
        assert rel_err(g[k], d["out"][k]) < 5 * TOL, k


@pytest.mark.parametrize("name", golden_case_names())
def test_eager_form_matches_reference(name):
    d = load_golden(name)
    p = O.Params.from_state_dict(d["params"])
    cfg = O.Config(use_sigmoid_mask=d["meta"]["use_sigmoid_mask"], tiny_thr=d["meta"]["tiny_thr"], eps=d["meta"]["eps"])
    y, g = O.reference_form_step(d["x"], d["mask"], p, cfg, d["gy"])
    assert rel_err(y, d["out"]["y"]) < 1e-6
    for k in GRADS:
        if g[k] is None:
            continue
        assert rel_err(g[k], d["out"][k]) < 1e-5, k


@pytest.mark.parametrize("name", ["base", "mixed_batch", "ties", "prob_mask", "nomask", "k3"])
def test_float64_ground_truth_brackets_fp32(name):
    """The float64 run of the explicit oracle is the ground truth used to budget GPU tolerances:
    the reference's own fp32 result must sit within 1e-5 of it."""
    d = load_golden(name)
    y64, g64, _ = _run_explicit(d, torch.float64)
    assert rel_err(d["out"]["y"], y64) < 1e-5
    assert rel_err(d["out"]["gx"], g64["gx"]) < 5e-5


def test_known_answers_from_survey(checksums):
    """SURVEY.md section 8c known answers (reference run, y.sum() loss) reproduced by the oracle at full 80x80."""
    big = checksums["big"]["survey_A1"]
    assert abs(big["y"]["sum"] - 616.109575) < 1e-3 and abs(big["gx"]["abs"] - 396785.619) < 0.5
    x, mask, _ = synth(2, 64, 80, 80)
    p = O.Params.default_init(64)
    y, ctx = O.forward(x, mask, p)
    g = O.backward(torch.ones_like(x), x, mask, p, O.Config(), ctx)
    assert abs(float(y.double().sum()) - big["y"]["sum"]) < 2e-3
    assert abs(float(y[0, 0, 0, 0]) - (-0.04846177)) < 1e-6
    for k in ("gx", "gmask", "gwsa", "gbeta"):
        got = checksum(g[k])
        assert abs(got["abs"] - big[k]["abs"]) <= 2e-5 * big[k]["abs"] + 1e-6, k
        assert abs(got["wsum"] - big[k]["wsum"]) <= 2e-5 * big[k]["abs"] + 1e-6, k


@pytest.mark.parametrize("name", ["cfg1_p3", "cfg1_p4", "cfg1_p5"])
def test_config1_checksums(checksums, name):
    """BASELINE.json configs[0] shapes (B=2): oracle vs reference checksums of y and every gradient."""
    ref = checksums["big"][name]
    B, C, H, W = ref["shape"]
    x, mask, gy = synth(B, C, H, W, mask_kind=ref["mask_kind"])
    p = O.Params.default_init(C)
    y, ctx = O.forward(x, mask, p)
    g = O.backward(gy, x, mask, p, O.Config(), ctx)
    got = dict(y=y, **{k: v for k, v in g.items() if v is not None})
    for k, v in got.items():
        c = checksum(v)
        scale = ref[k]["abs"] + 1e-12
        assert abs(c["sum"] - ref[k]["sum"]) <= 2e-5 * scale, k
        assert abs(c["wsum"] - ref[k]["wsum"]) <= 2e-5 * scale, k
        assert abs(c["abs"] - ref[k]["abs"]) <= 2e-5 * scale, k


def test_default_init_equals_reference_state_dict():
    d = load_golden("base")
    p = O.Params.default_init(64)
    for a, k in ((p.w1, "cam_mlp.0.weight"), (p.b1, "cam_mlp.0.bias"), (p.w2, "cam_mlp.2.weight"),
                 (p.b2, "cam_mlp.2.bias"), (p.wsa, "sam_conv.weight"), (p.beta, "beta")):
        assert torch.equal(a, d["params"][k]), k


def test_mask_shape_mismatch_is_an_error():
    x, _, _ = synth(1, 8, 6, 6)
    with pytest.raises(RuntimeError):
        O.forward(x, torch.zeros(1, 1, 3, 3), O.Params.default_init(8))


@pytest.mark.parametrize("out_size,in_size", [(80, 640), (40, 640), (20, 640), (68, 544), (34, 544), (17, 544),
                                              (7, 10), (10, 7), (33, 100), (1, 5), (5, 5), (13, 640)])
def test_nearest_index_bit_exact_vs_torch(out_size, in_size):
    """Integer index path of segmentation.py:103-110 == F.interpolate(mode='nearest') (third-party torch)."""
    import torch.nn.functional as F
    src = torch.arange(in_size, dtype=torch.float32).view(1, 1, 1, in_size)
    want = F.interpolate(src, size=(1, out_size), mode="nearest").view(-1).long()
    got = torch.from_numpy(O.nearest_src_index(out_size, in_size))
    assert torch.equal(got, want)
    t = torch.arange(in_size * in_size, dtype=torch.float32).view(1, 1, in_size, in_size)
    assert torch.equal(O.nearest_resize(t, out_size, out_size), F.interpolate(t, size=(out_size, out_size), mode="nearest"))
