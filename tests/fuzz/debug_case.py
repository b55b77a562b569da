"""Stage-wise comparison (forward statistics, then gradients) of specific shapes on the GPU box."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import rel_err, synth
from oracle import maskcbam_oracle as O
from mga_yolo_amd import functional as F

cases = [(3, 64, 14, 27, "mixed"), (9, 96, 25, 25, "randn"), (16, 48, 11, 6, "all_negative"), (16, 130, 23, 1, "sparse"), (16, 96, 3, 11, "none"),
         (3, 48, 17, 17, "mixed"), (5, 3, 23, 17, "mixed")]
for (B, C, H, W, kind) in cases:
    for k, r in itertools.product((7, 3, 1, 9), (16, 1)):
        x, mask, gy = synth(B, C, H, W, seed=5, mask_kind=kind)
        p = O.Params.default_init(C, r=r, k=k, seed=1)
        y_o, c = O.forward(x, mask, p)
        ps = [t.cuda() for t in (p.w1, p.b1, p.w2, p.b2, p.wsa, p.beta)]
        cfg = F.BlockConfig(hidden=p.w1.shape[0], k=k)
        try:
            y, v = F.forward_with_ctx(x.cuda(), None if mask is None else mask.cuda(), ps, cfg)
            torch.cuda.synchronize()
        except Exception as ex:
            print(B, C, H, W, kind, "k", k, "r", r, "ERROR", ex); continue
        N = H * W
        errs = dict(avg=rel_err(v["avg"], c.avg), mx=rel_err(v["mx"], c.mx), ca=rel_err(v["ca"], c.ca),
                    planes=rel_err(v["planes"], c.planes.reshape(B, 3, N)), cidx=float((v["cidx"].cpu().long() != c.cidx).sum()),
                    sa=rel_err(v["sa"], c.sa), y=rel_err(y, y_o))
        flag = "BAD" if any(e > 1e-4 for e in errs.values()) else "ok "
        print(flag, B, C, H, W, kind, "k", k, "r", r, {kk: f"{vv:.1e}" for kk, vv in errs.items()}, flush=True)
