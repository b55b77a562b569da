"""Autograd entry points of the mask-guided CBAM block on MI355X.

``mask_cbam(x, mask, ...)`` replaces the body of the reference's ``MaskCBAM.forward``
(mga_yolo/nn/modules/masked_cbam.py:154-171) for device tensors: forward and backward are each ONE call
into libmgacbam.so (include/mgacbam.h), which enqueues the hand-written gfx950 kernels on the current stream.
``mask_cbam_pyramid`` does the same for several independent pyramid levels (P3/P4/P5) in one call.

Device tensors never take any other path: a missing library raises (``_lib.LibraryMissing``).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class BlockConfig:
    """Non-tensor constructor state of one block (masked_cbam.py:34-50)."""
    hidden: int
    k: int = 7
    use_sigmoid_mask: bool = True
    tiny_thr: float = 1e-4
    eps: float = 1e-6


_DTYPES = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _raw_stream(dev: torch.device) -> int:
    """hipStream_t of torch's current stream on `dev` (the private accessor is ~10x cheaper than building a Stream object: this
    sits on the eager path's per-call critical path)."""
    try:
        return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    except AttributeError:
        return torch.cuda.current_stream(dev).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the common case costs one integer compare)."""
    __slots__ = ("dev", "guard")

    def __init__(self, dev):
        self.dev, self.guard = dev, None

    def __enter__(self):
        idx = self.dev.index
        if idx is not None and idx != torch.cuda.current_device():
            self.guard = torch.cuda.device(self.dev)
            self.guard.__enter__()

    def __exit__(self, *a):
        if self.guard is not None:
            self.guard.__exit__(*a)
_USE_PROJ = bool(int(os.environ.get("MGACBAM_PROJ", "0")))
# k_chan + k_apply as ONE x-resident launch (k_gate, MGACBAM_FWD_FUSE); MGACBAM_FUSE_FWD=0 restores the three-launch forward
_FUSE_FWD = bool(int(os.environ.get("MGACBAM_FUSE_FWD", "1")))
# transposed conv folded into the k_bwd_reduce1 launch (MGACBAM_BWD_FOLD); needs the zero-filled ctx tail the fused forward sets up
_FOLD_BWD = _FUSE_FWD and bool(int(os.environ.get("MGACBAM_FOLD_BWD", "1")))
# MGACBAM_CHECK_HANDOFF=1: synchronise after every library call and raise if an in-launch hand-off timed out (debugging switch;
# without it a time-out is still loud -- the tile is poisoned with NaN -- and handoff_report() reads every status word at once)
_CHECK_HANDOFF = bool(int(os.environ.get("MGACBAM_CHECK_HANDOFF", "0")))
SLOTS = 8  # tensors per level in the flat argument list: x, mask, w1, b1, w2, b2, wsa, beta
_FWD_STAGES = _lib.FWD_ALL | (_lib.FWD_FUSE if _FUSE_FWD else 0)
_BWD_STAGES = _lib.BWD_ALL | (_lib.BWD_FOLD if _FOLD_BWD else 0)


class HandoffTimeout(RuntimeError):
    """An in-launch hand-off (k_gate / folded k_bwd_reduce1) timed out: the device was shared with work that took the CUs its
    co-residency assumption needs.  The affected tiles were poisoned with NaN.  Set MGACBAM_FUSE_FWD=0 to run without hand-offs."""


class _CtxPool:
    """Saved-statistics buffers of the eager path, recycled per (device, stream, shape): the hand-off flags at the end of ctx are
    generation counters that are never reset, so a buffer is zero-filled ONCE when it is created and stays consistent over any
    number of calls -- the per-call zero-fill launch (and the allocation) disappear from the unchanged-trainer path."""

    def __init__(self, keep: int = 8):
        self.free, self.all, self.keep = {}, [], keep

    def take(self, key, nbytes: int, sync_off: int, dev):
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        buf[sync_off:].zero_()
        self.all.append((weakref.ref(buf), key))
        if len(self.all) > 4096:
            self.all = [(r, k) for r, k in self.all if r() is not None]
        return buf

    def give(self, key, buf):
        lst = self.free.setdefault(key, [])
        if len(lst) < self.keep:
            lst.append(buf)

    def clear(self):
        self.free.clear()
        self.all = [(r, k) for r, k in self.all if r() is not None]


_POOL = _CtxPool()


class _Lease:
    """Returns its buffer to the pool when the autograd context that owns it dies (after backward, or with the graph)."""
    __slots__ = ("key", "buf")

    def __init__(self, key, buf):
        self.key, self.buf = key, buf

    def __del__(self):
        try:
            _POOL.give(self.key, self.buf)
        except Exception:
            pass


def handoff_report(clear_pool: bool = False):
    """Synchronise and read the hand-off status word of every ctx buffer the eager path has handed out and that is still alive.
    Returns the number of buffers checked; raises HandoffTimeout if any word is set.  Call it where the training loop synchronises
    anyway (end of an epoch / validation)."""
    live = [(r(), k) for r, k in _POOL.all]
    live = [(b, k) for b, k in live if b is not None]
    if not live:
        return 0
    words = []
    for buf, key in live:
        B, Cc, H, W, hidden = key[2:7]
        off = _lib.ctx_layout(B, Cc, H, W, hidden)["status"]
        words.append(buf[off:off + 4].view(torch.int32))
    bad = [k for (b, k), w in zip(live, torch.cat(words).cpu().tolist()) if w != 0]
    if clear_pool:
        _POOL.clear()
    if bad:
        raise HandoffTimeout(f"in-launch hand-off timed out for shapes {sorted(set(k[2:6] for k in bad))}: affected tiles were "
                             "poisoned with NaN (is the GPU shared with other work?  MGACBAM_FUSE_FWD=0 runs without hand-offs)")
    return len(live)


def _check_status(bufs_and_shapes, what: str):
    for buf, (B, Cc, H, W, hidden) in bufs_and_shapes:
        off = _lib.ctx_layout(B, Cc, H, W, hidden)["status"]
        if int(buf[off:off + 4].view(torch.int32).item()) != 0:          # (.item() synchronises)
            raise HandoffTimeout(f"{what}: in-launch hand-off timed out for level (B,C,H,W)=({B},{Cc},{H},{W})")


def _aligned(t: torch.Tensor) -> torch.Tensor:
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def _ready(t: torch.Tensor) -> torch.Tensor:
    """Detached, contiguous, 16-byte aligned view/copy of t; the common case (already so) costs two attribute reads."""
    if t.is_contiguous() and t.data_ptr() % 16 == 0:
        return t                                               # (inside autograd.Function.forward: no graph is recorded)
    return _aligned(t.detach())


def _mask32(mask: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    m = mask
    if m.dim() == 3:
        m = m.unsqueeze(1)
    if m.dtype != torch.float32:
        m = m.float()
    return _ready(m)


def _check_level(x: torch.Tensor, mask: Optional[torch.Tensor], params: Sequence[torch.Tensor], cfg: BlockConfig):
    if x.dim() != 4:
        raise AssertionError("MaskCBAM expects a (B,C,H,W) feature")          # masked_cbam.py:160
    if x.dtype not in _DTYPES:
        raise TypeError(f"unsupported feature dtype {x.dtype}")
    B, Cc, H, W = x.shape
    if mask is not None:
        m = mask.unsqueeze(1) if mask.dim() == 3 else mask                    # masked_cbam.py:81-85
        if tuple(m.shape) != (B, 1, H, W):                                    # the reference's expand() raises too
            raise RuntimeError(f"mask shape {tuple(mask.shape)} does not match feature (B,1,H,W)=({B},1,{H},{W})")
    w1, b1, w2, b2, wsa, beta = params
    h, k = cfg.hidden, cfg.k
    want = {"w1": (h, Cc), "b1": (h,), "w2": (Cc, h), "b2": (Cc,), "wsa": (1, 3, k, k), "beta": ()}
    for name, t in zip(want, params):
        if tuple(t.shape) != want[name] or t.dtype != torch.float32 or t.device != x.device:
            raise ValueError(f"parameter {name}: expected fp32 {want[name]} on {x.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")


def _params_struct(params: Sequence[torch.Tensor], cfg: BlockConfig) -> _lib.Params:
    w1, b1, w2, b2, wsa, beta = params
    return _lib.Params(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), wsa.data_ptr(), beta.data_ptr(),
                       cfg.hidden, cfg.k, int(cfg.use_sigmoid_mask), cfg.tiny_thr, cfg.eps)


class _PyramidFn(torch.autograd.Function):
    """n independent levels; flat inputs = n x (x, mask|None, w1, b1, w2, b2, wsa, beta)."""

    @staticmethod
    def forward(ctx, cfgs: Tuple[BlockConfig, ...], *flat):
        n = len(cfgs)
        assert len(flat) == n * SLOTS and 1 <= n <= _lib.MAX_LEVELS
        lib = _lib.load()
        levels = (_lib.FwdLevel * n)()
        keep: List[Optional[torch.Tensor]] = []
        outs, meta, leases = [], [], []
        dev = flat[0].device
        if not flat[0].is_cuda:
            raise RuntimeError("mask_cbam: device tensors only (host tensors take the module's host path)")
        stream = _raw_stream(dev)
        for l in range(n):
            x, mask, *params = flat[l * SLOTS:(l + 1) * SLOTS]
            cfg = cfgs[l]
            if not x.is_cuda or x.device != dev:
                raise RuntimeError("mask_cbam: all features must live on the same GPU")
            _check_level(x, mask, params, cfg)
            B, Cc, H, W = x.shape
            xc = _ready(x)
            m32 = None if mask is None else _mask32(mask, B, H, W)
            pc = [_ready(p) for p in params]
            y = torch.empty_like(xc)
            # FWD_FUSE / BWD_FOLD contract: the hand-off flags at the end of ctx were zero-filled once (by the pool, at creation)
            # (the flags count calls PER TILE, so a buffer is only reused under the tiling it was used with.  The library derives every
            #  tiling from the level alone -- shape, element type, conv size k, knobs; never from the other levels of the call -- and
            #  exactly those are in the key)
            key = (dev.index, stream, B, Cc, H, W, cfg.hidden, x.dtype, _lib.ENV_EPOCH, cfg.k)
            lease = _Lease(key, _POOL.take(key, _lib.ctx_bytes(B, Cc, H, W, cfg.hidden), _lib.ctx_layout(B, Cc, H, W, cfg.hidden)["sync"], dev))
            cbuf = lease.buf
            leases.append(lease)
            L = levels[l]
            L.x, L.mask, L.y, L.ctx = xc.data_ptr(), (None if m32 is None else m32.data_ptr()), y.data_ptr(), cbuf.data_ptr()
            L.ctx_bytes = cbuf.numel()
            L.p = _params_struct(pc, cfg)
            L.B, L.C, L.H, L.W, L.dtype = B, Cc, H, W, _DTYPES[x.dtype]
            # opt-in (MGACBAM_PROJ=1): the backward of this call will want dL/dmask, let the forward save the W1-projection
            # planes for it.  Off by default: at YOLOv8n sizes what k_bwd_apply saves (x of P3) k_chan pays back
            proj = _USE_PROJ and mask is not None and mask.requires_grad and torch.is_grad_enabled()
            L.flags = _lib.FWD_SAVE_PROJ if proj else 0
            keep += [xc, m32, *pc]
            outs.append(y)
            meta.append(((None if mask is None else (mask.dtype, tuple(mask.shape))), proj))
        with _on_device(dev):
            rc = lib.mgacbam_forward_stages(levels, n, _FWD_STAGES, stream)
        if rc:
            _lib.check(rc, "mgacbam_forward_stages")
        if _CHECK_HANDOFF:
            _check_status([(ls.buf, ls.key[2:7]) for ls in leases], "mask_cbam forward")
        ctx.save_for_backward(*keep)
        ctx.cfgs, ctx.meta, ctx.leases = cfgs, meta, leases
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        cfgs, n = ctx.cfgs, len(ctx.cfgs)
        lib = _lib.load()
        saved = ctx.saved_tensors
        per = 2 + 6
        levels = (_lib.BwdLevel * n)()
        grads: List[Optional[torch.Tensor]] = [None]
        hold = []
        dev = saved[0].device
        for l in range(n):
            xc, m32, *pc = saved[l * per:(l + 1) * per]
            cbuf = ctx.leases[l].buf
            cfg = cfgs[l]
            B, Cc, H, W = xc.shape
            gy = gys[l]
            gy = torch.zeros_like(xc) if gy is None else _aligned(gy.to(xc.dtype))
            want_gmask = m32 is not None and ctx.needs_input_grad[1 + l * SLOTS + 1]
            gx = torch.empty_like(xc)
            gmask = torch.empty_like(m32) if want_gmask else None
            pg = [torch.empty_like(p) for p in pc]
            scratch = torch.empty(_lib.scratch_bytes(B, Cc, H, W, cfg.hidden, cfg.k), dtype=torch.uint8, device=dev)
            L = levels[l]
            L.x, L.mask, L.gy, L.ctx, L.scratch = (xc.data_ptr(), None if m32 is None else m32.data_ptr(), gy.data_ptr(),
                                                   cbuf.data_ptr(), scratch.data_ptr())
            L.ctx_bytes, L.scratch_bytes = cbuf.numel(), scratch.numel()
            L.gx, L.gmask = gx.data_ptr(), (None if gmask is None else gmask.data_ptr())
            L.gw1, L.gb1, L.gw2, L.gb2, L.gwsa, L.gbeta = (t.data_ptr() for t in pg)
            L.p = _params_struct(pc, cfg)
            L.B, L.C, L.H, L.W, L.dtype = B, Cc, H, W, _DTYPES[xc.dtype]
            L.flags = _lib.BWD_HAVE_PROJ if ctx.meta[l][1] else 0
            hold += [gy, scratch]
            grads += [gx, gmask, *pg]
        with _on_device(dev):
            rc = lib.mgacbam_backward_stages(levels, n, _BWD_STAGES, _raw_stream(dev))
        if rc:
            _lib.check(rc, "mgacbam_backward_stages")
        del hold
        for l in range(n):                                         # dL/dmask in the mask's own shape / element type -- AFTER the launch that
            gm = grads[1 + l * SLOTS + 1]                          # writes it (a cast enqueued before it would read unwritten memory: that
            if gm is not None:                                     # was the case for half-precision masks until the AMP test of round 3)
                mdtype, mshape = ctx.meta[l][0]
                grads[1 + l * SLOTS + 1] = gm.reshape(mshape).to(mdtype)
        if _CHECK_HANDOFF:
            _check_status([(ls.buf, ls.key[2:7]) for ls in ctx.leases], "mask_cbam backward")
        return tuple(grads)


def mask_cbam_pyramid(levels: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], Sequence[torch.Tensor], BlockConfig]]):
    """levels: [(x, mask|None, (w1,b1,w2,b2,wsa,beta), BlockConfig), ...] -> tuple of outputs (one library call)."""
    cfgs, flat = [], []
    for x, mask, params, cfg in levels:
        cfgs.append(cfg)
        flat += [x, mask, *params]
    return _PyramidFn.apply(tuple(cfgs), *flat)


def mask_cbam(x: torch.Tensor, mask: Optional[torch.Tensor], w1, b1, w2, b2, wsa, beta, cfg: BlockConfig) -> torch.Tensor:
    """y = x + softplus(beta) * (SAM(CAM(x, mask), mask) - x) for a device tensor x (B,C,H,W)."""
    return _PyramidFn.apply((cfg,), x, mask, w1, b1, w2, b2, wsa, beta)[0]


# ---------------------------------------------------------------------------------------------------------
# inspection helpers (tests / tooling): run the forward library call and view the saved statistics by name
# ---------------------------------------------------------------------------------------------------------
def forward_with_ctx(x, mask, params, cfg: BlockConfig, save_proj: bool = True):
    """-> (y, {name: tensor view into ctx}) without autograd; names follow mgacbam_ctx_layout_t."""
    lib = _lib.load()
    _check_level(x, mask, params, cfg)
    B, Cc, H, W = x.shape
    xc = _aligned(x.detach())
    m32 = None if mask is None else _aligned(mask.detach().reshape(B, 1, H, W).float())
    pc = [_aligned(p.detach()) for p in params]
    y = torch.empty_like(xc)
    cbuf = torch.zeros(_lib.ctx_bytes(B, Cc, H, W, cfg.hidden), dtype=torch.uint8, device=x.device)
    lv = (_lib.FwdLevel * 1)()
    L = lv[0]
    L.x, L.mask, L.y, L.ctx = xc.data_ptr(), (None if m32 is None else m32.data_ptr()), y.data_ptr(), cbuf.data_ptr()
    L.ctx_bytes = cbuf.numel()
    L.p = _params_struct(pc, cfg)
    L.B, L.C, L.H, L.W, L.dtype = B, Cc, H, W, _DTYPES[x.dtype]
    L.flags = _lib.FWD_SAVE_PROJ if (save_proj and mask is not None) else 0
    with torch.cuda.device(x.device):
        _lib.check(lib.mgacbam_forward(lv, 1, torch.cuda.current_stream(x.device).cuda_stream), "mgacbam_forward")
    return y, ctx_views(cbuf, B, Cc, H, W, cfg.hidden)


def ctx_views(cbuf: torch.Tensor, B, Cc, H, W, hidden) -> dict:
    lay = _lib.ctx_layout(B, Cc, H, W, hidden)
    HW = H * W
    shapes = dict(S=(B,), use=(B,), den=(B,), avg=(B, Cc), mx=(B, Cc), mavg=(B, Cc), valid=(B, Cc), amax=(B, Cc),
                  h_avg=(B, hidden), h_mx=(B, hidden), ca=(B, Cc), planes=(B, 3, HW), cidx=(B, HW), sa=(B, HW))
    if hidden <= _lib.PROJ_MAX_HIDDEN:
        shapes["proj"] = (B, hidden, HW)
    nflag = (HW + 15) // 16 + 1
    # hand-off state: (B, nflag) k_gate flags, [time-out status, 3 spare], (B) ca flags, (B, nflag) BWD_FOLD tile flags, (B, nflag) conv-tile flags,
    # then the merged backward launch's own (B, nflag) tile, conv-tile and dWsa-tile flags and (B, C) sweep flags
    shapes["sync"] = (6 * B * nflag + 4 + B + B * Cc,)
    ints = {"valid", "amax", "cidx", "sync"}
    out = {}
    for name, shp in shapes.items():
        n = 1
        for s in shp:
            n *= s
        raw = cbuf[lay[name]: lay[name] + 4 * n]
        out[name] = raw.view(torch.int32 if name in ints else torch.float32).reshape(shp)
    return out


# ---------------------------------------------------------------------------------------------------------
# MaskECA
# ---------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class EcaConfig:
    """Non-tensor state of a MaskECA block (masked_eca.py:57-65)."""
    k: int
    use_sigmoid_mask: bool = True
    tiny_thr: float = 1e-4
    eps: float = 1e-6


class _EcaFn(torch.autograd.Function):
    """n independent levels; flat inputs = n x (x, mask|None, conv1d.weight, beta)."""

    @staticmethod
    def forward(ctx, cfgs: Tuple[EcaConfig, ...], *flat):
        n = len(cfgs)
        assert len(flat) == 4 * n and 1 <= n <= _lib.MAX_LEVELS
        lib = _lib.load()
        levels = (_lib.EcaFwdLevel * n)()
        keep, outs, meta = [], [], []
        dev = flat[0].device
        for l in range(n):
            x, mask, w, beta = flat[4 * l:4 * l + 4]
            cfg = cfgs[l]
            if not x.is_cuda or x.device != dev:
                raise RuntimeError("mask_eca: all features must live on the same GPU")
            if x.dim() != 4:
                raise AssertionError("feature must be (B,C,H,W)")                    # masked_eca.py:174
            if x.dtype not in _DTYPES:
                raise TypeError(f"unsupported feature dtype {x.dtype}")
            B, Cc, H, W = x.shape
            if mask is not None:
                m4 = mask.unsqueeze(1) if mask.dim() == 3 else mask
                if tuple(m4.shape) != (B, 1, H, W):
                    raise RuntimeError(f"mask shape {tuple(mask.shape)} does not match feature (B,1,H,W)=({B},1,{H},{W})")
            if tuple(w.shape) != (1, 1, cfg.k) or beta.dim() != 0:
                raise ValueError(f"MaskECA parameters: expected conv1d.weight (1,1,{cfg.k}) and scalar beta")
            xc = _aligned(x.detach())
            m32 = None if mask is None else _aligned(mask.detach().reshape(B, 1, H, W).float())
            wc, bc = _aligned(w.detach().float()), _aligned(beta.detach().float())
            y = torch.empty_like(xc)
            cbuf = torch.empty(lib.mgacbam_eca_ctx_bytes(B, Cc, H, W), dtype=torch.uint8, device=dev)
            L = levels[l]
            L.x, L.mask, L.y, L.ctx = xc.data_ptr(), (None if m32 is None else m32.data_ptr()), y.data_ptr(), cbuf.data_ptr()
            L.ctx_bytes = cbuf.numel()
            L.p = _lib.EcaParams(wc.data_ptr(), bc.data_ptr(), cfg.k, int(cfg.use_sigmoid_mask), cfg.tiny_thr, cfg.eps)
            L.B, L.C, L.H, L.W, L.dtype = B, Cc, H, W, _DTYPES[x.dtype]
            keep += [xc, m32, cbuf, wc, bc]
            outs.append(y)
            meta.append(None if mask is None else (mask.dtype, tuple(mask.shape)))
        with torch.cuda.device(dev):
            _lib.check(lib.mgacbam_eca_forward(levels, n, torch.cuda.current_stream(dev).cuda_stream), "mgacbam_eca_forward")
        ctx.save_for_backward(*keep)
        ctx.cfgs, ctx.meta = cfgs, meta
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        cfgs, n = ctx.cfgs, len(ctx.cfgs)
        lib = _lib.load()
        saved = ctx.saved_tensors
        levels = (_lib.EcaBwdLevel * n)()
        grads: List[Optional[torch.Tensor]] = [None]
        hold = []
        dev = saved[0].device
        for l in range(n):
            xc, m32, cbuf, wc, bc = saved[5 * l:5 * l + 5]
            cfg = cfgs[l]
            B, Cc, H, W = xc.shape
            gy = gys[l]
            gy = torch.zeros_like(xc) if gy is None else _aligned(gy.to(xc.dtype))
            want_gmask = m32 is not None and ctx.needs_input_grad[1 + 4 * l + 1]
            gx = torch.empty_like(xc)
            gmask = torch.empty_like(m32) if want_gmask else None
            gw, gb = torch.empty_like(wc), torch.empty_like(bc)
            scratch = torch.empty(lib.mgacbam_eca_scratch_bytes(B, Cc, H, W), dtype=torch.uint8, device=dev)
            L = levels[l]
            L.x, L.mask, L.gy, L.ctx, L.scratch = (xc.data_ptr(), None if m32 is None else m32.data_ptr(), gy.data_ptr(),
                                                   cbuf.data_ptr(), scratch.data_ptr())
            L.ctx_bytes, L.scratch_bytes = cbuf.numel(), scratch.numel()
            L.gx, L.gmask, L.gw, L.gbeta = gx.data_ptr(), (None if gmask is None else gmask.data_ptr()), gw.data_ptr(), gb.data_ptr()
            L.p = _lib.EcaParams(wc.data_ptr(), bc.data_ptr(), cfg.k, int(cfg.use_sigmoid_mask), cfg.tiny_thr, cfg.eps)
            L.B, L.C, L.H, L.W, L.dtype = B, Cc, H, W, _DTYPES[xc.dtype]
            hold += [gy, scratch]
            grads += [gx, gmask, gw, gb]
        with torch.cuda.device(dev):
            _lib.check(lib.mgacbam_eca_backward(levels, n, torch.cuda.current_stream(dev).cuda_stream), "mgacbam_eca_backward")
        del hold
        for l in range(n):                                         # (after the launch that writes it, see _PyramidFn.backward)
            gm = grads[1 + 4 * l + 1]
            if gm is not None:
                mdtype, mshape = ctx.meta[l]
                grads[1 + 4 * l + 1] = gm.reshape(mshape).to(mdtype)
        return tuple(grads)


def mask_eca(x: torch.Tensor, mask: Optional[torch.Tensor], w: torch.Tensor, beta: torch.Tensor, cfg: EcaConfig) -> torch.Tensor:
    """y = x * (1 + softplus(beta) * (sigmoid(conv1d(masked_avg(x, mask))) - 0.5)) for a device tensor (masked_eca.py:167-196)."""
    return _EcaFn.apply((cfg,), x, mask, w, beta)[0]


def mask_eca_pyramid(levels: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor, torch.Tensor, EcaConfig]]):
    cfgs, flat = [], []
    for x, mask, w, beta, cfg in levels:
        cfgs.append(cfg)
        flat += [x, mask, w, beta]
    return _EcaFn.apply(tuple(cfgs), *flat)


def resize_nearest(src: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """F.interpolate(src, (out_h,out_w), mode='nearest') for fp32 (...,H,W) device tensors: the integer index path of
    mga_yolo/nn/losses/segmentation.py:103-110, bit-exact (pure gather)."""
    if not src.is_cuda or src.dtype != torch.float32 or src.dim() < 2:
        raise TypeError("resize_nearest expects an fp32 device tensor (..., H, W)")
    lib = _lib.load()
    s = _aligned(src)
    in_h, in_w = s.shape[-2:]
    planes = s.numel() // (in_h * in_w)
    dst = torch.empty(*s.shape[:-2], out_h, out_w, dtype=torch.float32, device=s.device)
    with torch.cuda.device(s.device):
        _lib.check(lib.mgacbam_resize_nearest(s.data_ptr(), dst.data_ptr(), planes, in_h, in_w, out_h, out_w,
                                              torch.cuda.current_stream(s.device).cuda_stream), "mgacbam_resize_nearest")
    return dst


# ---------------------------------------------------------------------------------------------------------
# ProbMaskGater: gumbel / hard_st sampling as one launch; the uniforms come from the caller
# ---------------------------------------------------------------------------------------------------------
class _GaterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, u1, u2, tau: float, p_min: float, threshold: float, hard: bool):
        lib = _lib.load()
        pc, a, b = (_ready(t) for t in (p, u1, u2))
        if pc.dtype != torch.float32 or a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != pc.shape or b.shape != pc.shape:
            raise RuntimeError("prob_mask_gate: p, u1, u2 must be fp32 tensors of one shape")
        out, msoft = torch.empty_like(pc), torch.empty_like(pc)
        cfg = _lib.PmgCfg(float(tau), float(p_min), float(threshold), int(bool(hard)))
        with torch.cuda.device(pc.device):
            _lib.check(lib.mgapmg_forward(pc.data_ptr(), a.data_ptr(), b.data_ptr(), out.data_ptr(), msoft.data_ptr(), pc.numel(),
                                          C.byref(cfg), torch.cuda.current_stream(pc.device).cuda_stream), "mgapmg_forward")
        ctx.save_for_backward(pc, msoft)
        ctx.cfg = (float(tau), float(p_min), float(threshold), int(bool(hard)))
        return out

    @staticmethod
    def backward(ctx, gout):
        pc, msoft = ctx.saved_tensors
        lib = _lib.load()
        g = _ready(gout.to(torch.float32))
        gp = torch.empty_like(pc)
        cfg = _lib.PmgCfg(*ctx.cfg)
        with torch.cuda.device(pc.device):
            _lib.check(lib.mgapmg_backward(pc.data_ptr(), msoft.data_ptr(), g.data_ptr(), gp.data_ptr(), pc.numel(), C.byref(cfg),
                                           torch.cuda.current_stream(pc.device).cuda_stream), "mgapmg_backward")
        return gp, None, None, None, None, None, None


def prob_mask_gate(p: torch.Tensor, u1: torch.Tensor, u2: torch.Tensor, tau: float = 1.0, p_min: float = 0.0, threshold: float = 0.5,
                   hard: bool = False) -> torch.Tensor:
    """Gumbel-sigmoid gate of ProbMaskGater for device tensors: max(clamp(p,0,1), p_min) -> sigmoid((logit + logistic(u1,u2)) / tau),
    thresholded with a straight-through gradient when ``hard``.  u1, u2: uniform draws, as torch.rand gives them."""
    return _GaterFn.apply(p, u1, u2, tau, p_min, threshold, hard)


# ---------------------------------------------------------------------------------------------------------
# MGAMaskHead: Conv1x1 -> BatchNorm2d -> SiLU -> Conv3x3 as 3 launches forward, 5 backward (csrc/head.cuh)
# ---------------------------------------------------------------------------------------------------------
def _head_params(w1, gamma, beta, rmean, rvar, nbt, wh, bh, hidden, eps, momentum, training) -> "_lib.HeadParams":
    return _lib.HeadParams(w1.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(),
                           None if nbt is None else nbt.data_ptr(), wh.data_ptr(), bh.data_ptr(), hidden, eps, momentum, int(training))


class _HeadFn(torch.autograd.Function):
    """n independent levels; flat inputs = n x (x, proj.0.weight, proj.1.weight, proj.1.bias, head.weight, head.bias);
    ``state`` = per level (running_mean, running_var, num_batches_tracked | None, eps, momentum, training): buffers updated in place."""

    @staticmethod
    def forward(ctx, state: tuple, *flat):
        n = len(state)
        assert len(flat) == 6 * n and 1 <= n <= _lib.MAX_LEVELS
        lib = _lib.load()
        levels = (_lib.HeadFwdLevel * n)()
        keep, outs, meta = [], [], []
        dev = flat[0].device
        if not flat[0].is_cuda:
            raise RuntimeError("mask_head: device tensors only (host tensors take the module's host path)")
        for l in range(n):
            x, w1, gamma, beta, wh, bh = flat[6 * l:6 * l + 6]
            rmean, rvar, nbt, eps, momentum, training = state[l]
            if x.dim() != 4 or x.dtype not in _DTYPES or x.device != dev:
                raise RuntimeError("mask_head: x must be a (B,C,H,W) fp32 / fp16 / bf16 tensor on one GPU")
            B, Cc, H, W = x.shape
            hid = w1.shape[0]
            if tuple(w1.shape[:2]) != (hid, Cc) or tuple(wh.shape) != (1, hid, 3, 3) or bh.numel() != 1 or gamma.numel() != hid:
                raise ValueError(f"mask_head: parameter shapes do not match C={Cc}, hidden={hid} (out_channels must be 1)")
            if training and B * H * W == 1:                     # torch.nn.functional.batch_norm refuses this too (no variance from one value)
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
            xc = _ready(x)
            pc = [_ready(t.float() if t.dtype != torch.float32 else t) for t in (w1, gamma, beta, wh, bh)]
            for t in (rmean, rvar):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                    raise ValueError("mask_head: running statistics must be contiguous fp32 tensors on the feature's device")
            logits = torch.empty(B, 1, H, W, dtype=x.dtype, device=dev)
            cbuf = torch.empty(lib.mgahead_ctx_bytes(B, Cc, H, W, hid), dtype=torch.uint8, device=dev)
            L = levels[l]
            L.x, L.logits, L.ctx, L.ctx_bytes = xc.data_ptr(), logits.data_ptr(), cbuf.data_ptr(), cbuf.numel()
            L.p = _head_params(*pc[:3], rmean, rvar, nbt, *pc[3:], hid, float(eps), float(momentum), training)
            L.B, L.C, L.H, L.W, L.dtype, L.flags = B, Cc, H, W, _DTYPES[x.dtype], 0
            keep += [xc, cbuf, *pc, rmean, rvar]
            outs.append(logits)
            meta.append((hid, float(eps), float(momentum), bool(training), tuple(w1.shape)))
        with torch.cuda.device(dev):
            _lib.check(lib.mgahead_forward(levels, n, torch.cuda.current_stream(dev).cuda_stream), "mgahead_forward")
        ctx.save_for_backward(*keep)
        ctx.meta = meta
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gls):
        n = len(ctx.meta)
        lib = _lib.load()
        saved = ctx.saved_tensors
        levels = (_lib.HeadBwdLevel * n)()
        grads: List[Optional[torch.Tensor]] = [None]
        hold = []
        dev = saved[0].device
        for l in range(n):
            xc, cbuf, w1, gamma, beta, wh, bh, rmean, rvar = saved[9 * l:9 * l + 9]
            hid, eps, momentum, training, w1_shape = ctx.meta[l]
            B, Cc, H, W = xc.shape
            gl = gls[l]
            gl = torch.zeros(B, 1, H, W, dtype=xc.dtype, device=dev) if gl is None else _aligned(gl.to(xc.dtype))
            gx = torch.empty_like(xc)
            pg = [torch.empty_like(t) for t in (w1, gamma, beta, wh, bh)]
            scratch = torch.empty(lib.mgahead_bwd_scratch_bytes(B, Cc, H, W, hid), dtype=torch.uint8, device=dev)
            L = levels[l]
            L.x, L.g_logits, L.g_logits2, L.ctx, L.scratch, L.gx = xc.data_ptr(), gl.data_ptr(), None, cbuf.data_ptr(), scratch.data_ptr(), gx.data_ptr()
            L.ctx_bytes, L.scratch_bytes = cbuf.numel(), scratch.numel()
            L.gw1, L.gbn_weight, L.gbn_bias, L.gwh, L.gbh = (t.data_ptr() for t in pg)
            L.p = _head_params(w1, gamma, beta, rmean, rvar, None, wh, bh, hid, eps, momentum, training)
            L.B, L.C, L.H, L.W, L.dtype, L.flags = B, Cc, H, W, _DTYPES[xc.dtype], 0
            hold += [gl, scratch]
            pg[0] = pg[0].view(w1_shape)
            grads += [gx, *pg]
        with torch.cuda.device(dev):
            _lib.check(lib.mgahead_backward(levels, n, torch.cuda.current_stream(dev).cuda_stream), "mgahead_backward")
        del hold
        return tuple(grads)


def mask_head(x: torch.Tensor, w1, bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, wh, bh,
              eps: float = 1e-5, momentum: float = 0.1, training: bool = True) -> torch.Tensor:
    """Mask logits (B,1,H,W) = Conv3x3(SiLU(BatchNorm2d(Conv1x1(x)))) for a device tensor x; in training the running statistics
    (and num_batches_tracked) are updated in place exactly as torch's BatchNorm2d does (mga_yolo/nn/modules/segmentation.py:56-110)."""
    return _HeadFn.apply(((running_mean, running_var, num_batches_tracked, eps, momentum, training),), x, w1, bn_weight, bn_bias, wh, bh)[0]


def mask_head_pyramid(levels):
    """levels: [(x, w1, bn_weight, bn_bias, running_mean, running_var, num_batches_tracked, wh, bh, eps, momentum, training), ...]
    -> tuple of logits; ONE library call each way for all levels (the three heads read different features: they are independent)."""
    state, flat = [], []
    for x, w1, g_, b_, rm, rv, nbt, wh, bh, eps, mom, tr in levels:
        state.append((rm, rv, nbt, eps, mom, tr))
        flat += [x, w1, g_, b_, wh, bh]
    return _HeadFn.apply(tuple(state), *flat)
