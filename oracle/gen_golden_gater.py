"""Golden vectors for ProbMaskGater from the REFERENCE module itself, run once in the build container:

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python oracle/gen_golden_gater.py

mga_yolo/nn/modules/probmaskgater.py imports only torch, so it imports cleanly here.  With `seed` set the module draws its
uniforms from torch.Generator(seed + call counter) (probmaskgater.py:43-49, 53-56), so the same two draws can be repeated outside
it: each case stores the input, the two uniform tensors the module consumed, its output and the input gradient for a random
upstream gradient.  Data only -- no reference source is copied."""
import os
import sys

import numpy as np
import torch

from mga_yolo.nn.modules.probmaskgater import ProbMaskGater   # the reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = [  # name, mode, tau, p_min, threshold, shape, seed
    ("gater_gumbel", "gumbel", 1.0, 0.0, 0.5, (2, 1, 12, 10), 7),
    ("gater_gumbel_tau", "gumbel", 0.5, 0.2, 0.5, (3, 1, 9, 9), 11),
    ("gater_hard", "hard_st", 1.0, 0.0, 0.5, (2, 1, 8, 16), 3),
    ("gater_hard_thr", "hard_st", 0.7, 0.1, 0.35, (1, 1, 17, 5), 5),
    ("gater_3d", "gumbel", 1.0, 0.0, 0.5, (2, 6, 7), 9),            # (N,H,W) input
]
for name, mode, tau, p_min, thr, shape, seed in CASES:
    g = torch.Generator().manual_seed(100 + seed)
    p = torch.rand(shape, generator=g) * 1.6 - 0.3                  # also values outside [0,1]: the clamp and its gradient gate
    p.view(-1)[:4] = torch.tensor([0.0, 1.0, 1e-7, 1.0 - 1e-7])     # clamp edges of the logit
    m = ProbMaskGater(mode=mode, tau=tau, p_min=p_min, threshold=thr, seed=seed)
    m.train()
    x = p.clone().requires_grad_(True)
    out = m(x)
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout)
    shape4 = out.shape
    gen = torch.Generator().manual_seed(seed + 0)                   # the module's first call: counter 0
    u1 = torch.rand(shape4, dtype=torch.float32, generator=gen)
    u2 = torch.rand(shape4, dtype=torch.float32, generator=gen)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), p=p.numpy(), u1=u1.numpy(), u2=u2.numpy(), out=out.detach().numpy(),
                        gout=gout.numpy(), gp=x.grad.numpy(), mode=mode, tau=tau, p_min=p_min, threshold=thr, seed=seed)
    print(name, tuple(out.shape), float(out.sum()))
