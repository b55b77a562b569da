"""Static executor for the block over a feature pyramid: pre-allocated buffers, pre-built C-ABI level tables and
hipGraph capture of the step.  This is the launch path for steady-state training loops on MI355X -- the YOLOv8n shapes
are launch-latency bound (a full pass over the 52 MB P3 feature is ~10 us of HBM time), so the ~30 kernel launches of a
step are recorded once and replayed, instead of being issued from Python every step.

A ``PyramidPlan`` owns, per level: x, mask, y, gy, gx, gmask, ctx, scratch; and ONE flat fp32 bucket holding the parameter
gradients of all levels (what data-parallel training all-reduces, see ``dp.py``).  ``forward`` / ``backward`` each make ONE
library call on the current stream (5 kernel launches per step in total).  ``backward_params`` + ``backward_inputs`` is
the split form for callers that want the parameter gradients early (see ``dp.py``).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import os

import torch

from . import _lib
from .functional import BlockConfig, HandoffTimeout, _DTYPES, _params_struct, ctx_views

PARAM_NAMES = ("w1", "b1", "w2", "b2", "wsa", "beta")


class PyramidPlan:
    def __init__(self, shapes: Sequence[Tuple[int, int, int, int]], params: Sequence[Sequence[torch.Tensor]],
                 cfgs: Sequence[BlockConfig], dtype: torch.dtype = torch.float32, device="cuda",
                 with_mask: bool = True, want_gmask: bool = True, use_proj: bool = False,
                 fuse_forward: Optional[bool] = None, grad_bucket: Optional[torch.Tensor] = None):
        # fuse_forward: k_chan + k_apply as ONE x-resident launch, k_gate (MGACBAM_FWD_FUSE); None = env MGACBAM_FUSE_FWD (on)
        self.fuse_forward = bool(int(os.environ.get("MGACBAM_FUSE_FWD", "1"))) if fuse_forward is None else bool(fuse_forward)
        # transposed conv folded into the k_bwd_reduce1 launch (MGACBAM_BWD_FOLD) whenever the whole backward is one call
        self.fold_backward = bool(int(os.environ.get("MGACBAM_FOLD_BWD", "1")))
        assert len(shapes) == len(params) == len(cfgs) and 1 <= len(shapes) <= _lib.MAX_LEVELS
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n = len(shapes)
        self.shapes, self.cfgs, self.dtype = list(shapes), list(cfgs), dtype
        self.params = [[p.detach().to(self.device, torch.float32).contiguous() for p in ps] for ps in params]
        dev = self.device
        self.x, self.mask, self.y, self.gy, self.gx, self.gmask, self.ctx, self.scratch = ([] for _ in range(8))
        n_grad = sum(p.numel() for ps in self.params for p in ps)
        if grad_bucket is not None:                              # a slice of a larger bucket owned by the caller (slice.SlicePlan)
            assert grad_bucket.numel() == n_grad and grad_bucket.dtype == torch.float32 and grad_bucket.is_contiguous()
            self.grad_bucket = grad_bucket
        else:
            self.grad_bucket = torch.zeros(n_grad, dtype=torch.float32, device=dev)
        self.param_grads: List[List[torch.Tensor]] = []
        self._fwd = (_lib.FwdLevel * self.n)()
        self._bwd = (_lib.BwdLevel * self.n)()
        off = 0
        for l, ((B, C, H, W), ps, cfg) in enumerate(zip(shapes, self.params, cfgs)):
            mk = lambda *s, dt=dtype: torch.zeros(*s, dtype=dt, device=dev)
            self.x.append(mk(B, C, H, W)); self.y.append(mk(B, C, H, W)); self.gy.append(mk(B, C, H, W)); self.gx.append(mk(B, C, H, W))
            self.mask.append(mk(B, 1, H, W, dt=torch.float32) if with_mask else None)
            self.gmask.append(mk(B, 1, H, W, dt=torch.float32) if (with_mask and want_gmask) else None)
            self.ctx.append(torch.zeros(_lib.ctx_bytes(B, C, H, W, cfg.hidden), dtype=torch.uint8, device=dev))
            self.scratch.append(torch.zeros(_lib.scratch_bytes(B, C, H, W, cfg.hidden, cfg.k), dtype=torch.uint8, device=dev))
            views = []
            for p in ps:
                views.append(self.grad_bucket[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self.param_grads.append(views)
            ptr = lambda t: None if t is None else t.data_ptr()
            F, Bw = self._fwd[l], self._bwd[l]
            F.x, F.mask, F.y, F.ctx = ptr(self.x[l]), ptr(self.mask[l]), ptr(self.y[l]), ptr(self.ctx[l])
            F.ctx_bytes = self.ctx[l].numel()
            F.p = _params_struct(ps, cfg)
            F.B, F.C, F.H, F.W, F.dtype = B, C, H, W, _DTYPES[dtype]
            F.flags = _lib.FWD_SAVE_PROJ if (with_mask and want_gmask and use_proj) else 0
            Bw.x, Bw.mask, Bw.gy, Bw.ctx, Bw.scratch = ptr(self.x[l]), ptr(self.mask[l]), ptr(self.gy[l]), ptr(self.ctx[l]), ptr(self.scratch[l])
            Bw.ctx_bytes, Bw.scratch_bytes = self.ctx[l].numel(), self.scratch[l].numel()
            Bw.gx, Bw.gmask = ptr(self.gx[l]), ptr(self.gmask[l])
            Bw.gw1, Bw.gb1, Bw.gw2, Bw.gb2, Bw.gwsa, Bw.gbeta = (v.data_ptr() for v in views)
            Bw.p = _params_struct(ps, cfg)
            Bw.B, Bw.C, Bw.H, Bw.W, Bw.dtype = B, C, H, W, _DTYPES[dtype]
            Bw.flags = _lib.BWD_HAVE_PROJ if (with_mask and want_gmask and use_proj) else 0

    # ------------------------------------------------------------------ library calls on the current stream
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self, stages: int = _lib.FWD_ALL):
        both = _lib.FWD_STAGES["chan"] | _lib.FWD_STAGES["apply"]
        if self.fuse_forward and (stages & both) == both:
            stages |= _lib.FWD_FUSE     # ctx is zero-filled at allocation, as the flag's contract asks
        _lib.check(self.lib.mgacbam_forward_stages(self._fwd, self.n, stages, self._stream()), "mgacbam_forward_stages")

    def check_handoff(self) -> None:
        """Synchronise and raise HandoffTimeout if any in-launch hand-off of any call on this plan timed out (status word of each
        level's ctx, include/mgacbam.h).  Callers invoke it where they synchronise anyway: bench.py after the timed region,
        training loops after a batch of graph replays."""
        torch.cuda.synchronize(self.device)
        words = []
        for l, (B, C, H, W) in enumerate(self.shapes):
            off = _lib.ctx_layout(B, C, H, W, self.cfgs[l].hidden)["status"]
            words.append(self.ctx[l][off:off + 4].view(torch.int32))
        bad = [self.shapes[l] for l, w in enumerate(torch.cat(words).cpu().tolist()) if w != 0]
        if bad:
            raise HandoffTimeout(f"in-launch hand-off timed out for levels {bad}: affected tiles were poisoned with NaN "
                                 "(is the GPU shared with other work?  MGACBAM_FUSE_FWD=0 runs without hand-offs)")

    def gate_active(self) -> bool:
        """True when the last fused forward really ran k_gate (the library falls back to k_chan + k_apply for groups with a
        level whose shape is not eligible): k_gate bumps the hand-off flags at the end of ctx, the fallback never touches them."""
        if not self.fuse_forward:
            return False
        torch.cuda.synchronize(self.device)
        return all(int(self.ctx_view(l)["sync"][:Bn * ((H * W + 15) // 16 + 1)].max()) != 0 for l, (Bn, C, H, W) in enumerate(self.shapes))

    def fold_active(self) -> bool:
        """True when MGACBAM_BWD_FOLD really folds the transposed conv into the k_bwd_reduce1 launch for these shapes (the library
        falls back to two launches for ineligible groups): the folded launch bumps the backward hand-off counters."""
        if not self.fold_backward:
            return False
        B = _lib.BWD_STAGES

        def counters():
            torch.cuda.synchronize(self.device)
            out = []
            for l, (Bn, C, H, W) in enumerate(self.shapes):
                nf = Bn * ((H * W + 15) // 16 + 1)
                out.append(self.ctx_view(l)["sync"][nf + 4 + Bn:nf + 4 + Bn + nf].clone())
            return out
        self.forward()
        before = counters()
        self.backward(B["reduce1"] | B["convT"] | _lib.BWD_FOLD)
        after = counters()
        self.backward(_lib.BWD_ALL & ~(B["reduce1"] | B["convT"]))      # finishes the step
        torch.cuda.synchronize(self.device)
        return all(not torch.equal(a, b) for a, b in zip(before, after))

    def backward(self, stages: int = _lib.BWD_ALL):
        if stages == _lib.BWD_ALL and self.fold_backward:
            stages |= _lib.BWD_FOLD     # ctx is zero-filled at allocation, as the flag's contract asks
        _lib.check(self.lib.mgacbam_backward_stages(self._bwd, self.n, stages, self._stream()), "mgacbam_backward_stages")

    def backward_params(self):
        """Every stage the parameter gradients depend on, as separate launches: they are complete when this returns to the
        stream, so a data-parallel caller can start their all-reduce and overlap it with ``backward_inputs``."""
        self.backward(_lib.BWD_PARAMS)

    def backward_inputs(self):
        """k_bwd_apply: gx and gmask."""
        self.backward(_lib.BWD_INPUTS)

    # ------------------------------------------------------------------ hipGraph capture
    def capture(self, fn) -> "torch.cuda.CUDAGraph":
        """Record ``fn()`` (library calls on this plan) into a graph; warm up on a side stream first, as capture requires."""
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        return g

    # ------------------------------------------------------------------ sizes / inspection
    def elements(self) -> int:
        """E = sum over levels of B*C*H*W."""
        return sum(B * C * H * W for B, C, H, W in self.shapes)

    def images(self) -> int:
        return self.shapes[0][0]

    def ctx_view(self, level: int) -> dict:
        B, C, H, W = self.shapes[level]
        return ctx_views(self.ctx[level], B, C, H, W, self.cfgs[level].hidden)

    def named_param_grads(self, level: int) -> dict:
        return dict(zip(("g" + n for n in PARAM_NAMES), self.param_grads[level]))


class EcaPyramidPlan:
    """The same static executor for MaskECA: per level x, mask, y, gy, gx, gmask, ctx, scratch and one flat bucket
    with the parameter gradients (conv1d.weight, beta) of all levels; ``forward`` / ``backward`` = one library call each
    (2 kernel launches each for all levels together)."""

    def __init__(self, shapes, params, cfgs, dtype=torch.float32, device="cuda", with_mask=True, want_gmask=True):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n, self.shapes, self.cfgs, self.dtype = len(shapes), list(shapes), list(cfgs), dtype
        dev = self.device
        self.params = [[p.detach().to(dev, torch.float32).contiguous() for p in ps] for ps in params]
        self.x, self.mask, self.y, self.gy, self.gx, self.gmask, self.ctx, self.scratch = ([] for _ in range(8))
        self.grad_bucket = torch.zeros(sum(p.numel() for ps in self.params for p in ps), dtype=torch.float32, device=dev)
        self.param_grads = []
        self._fwd, self._bwd = (_lib.EcaFwdLevel * self.n)(), (_lib.EcaBwdLevel * self.n)()
        off = 0
        ptr = lambda t: None if t is None else t.data_ptr()
        for l, ((B, C, H, W), (w, beta), cfg) in enumerate(zip(shapes, self.params, cfgs)):
            mk = lambda *s_, dt=dtype: torch.zeros(*s_, dtype=dt, device=dev)
            self.x.append(mk(B, C, H, W)); self.y.append(mk(B, C, H, W)); self.gy.append(mk(B, C, H, W)); self.gx.append(mk(B, C, H, W))
            self.mask.append(mk(B, 1, H, W, dt=torch.float32) if with_mask else None)
            self.gmask.append(mk(B, 1, H, W, dt=torch.float32) if (with_mask and want_gmask) else None)
            self.ctx.append(torch.zeros(self.lib.mgacbam_eca_ctx_bytes(B, C, H, W), dtype=torch.uint8, device=dev))
            self.scratch.append(torch.zeros(self.lib.mgacbam_eca_scratch_bytes(B, C, H, W), dtype=torch.uint8, device=dev))
            gw = self.grad_bucket[off:off + w.numel()].view(w.shape); off += w.numel()
            gb = self.grad_bucket[off:off + 1].view(()); off += 1
            self.param_grads.append([gw, gb])
            P = _lib.EcaParams(w.data_ptr(), beta.data_ptr(), cfg.k, int(cfg.use_sigmoid_mask), cfg.tiny_thr, cfg.eps)
            F, Bw = self._fwd[l], self._bwd[l]
            F.x, F.mask, F.y, F.ctx, F.p = ptr(self.x[l]), ptr(self.mask[l]), ptr(self.y[l]), ptr(self.ctx[l]), P
            F.ctx_bytes = Bw.ctx_bytes = self.ctx[l].numel()
            Bw.scratch_bytes = self.scratch[l].numel()
            F.B, F.C, F.H, F.W, F.dtype = B, C, H, W, _DTYPES[dtype]
            Bw.x, Bw.mask, Bw.gy, Bw.ctx, Bw.scratch = ptr(self.x[l]), ptr(self.mask[l]), ptr(self.gy[l]), ptr(self.ctx[l]), ptr(self.scratch[l])
            Bw.gx, Bw.gmask, Bw.gw, Bw.gbeta, Bw.p = ptr(self.gx[l]), ptr(self.gmask[l]), gw.data_ptr(), gb.data_ptr(), P
            Bw.B, Bw.C, Bw.H, Bw.W, Bw.dtype = B, C, H, W, _DTYPES[dtype]

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self):
        _lib.check(self.lib.mgacbam_eca_forward(self._fwd, self.n, self._stream()), "mgacbam_eca_forward")

    def backward(self):
        _lib.check(self.lib.mgacbam_eca_backward(self._bwd, self.n, self._stream()), "mgacbam_eca_backward")

    capture = PyramidPlan.capture

    def elements(self) -> int:
        return sum(B * C * H * W for B, C, H, W in self.shapes)
