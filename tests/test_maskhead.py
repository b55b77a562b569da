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


def _close(got, want, tol, what):
    # relative to the tensor's scale, with an absolute floor for all-zero expectations (stride probe: logits are exactly the bias)
    scale = float(want.double().abs().max())
    err = float((got.double() - want.double()).abs().max())
    assert err <= tol * max(scale, 1e-6), (what, err, scale)


@pytest.mark.parametrize("name", head_golden_names())
def test_oracle_matches_the_reference_golden(name):
    d = load_head_golden(name)
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
