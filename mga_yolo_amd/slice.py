"""Static executor of the LAYER-LOOP SLICE of one training step around the hot path (SURVEY 8d images/s definition (2), 8f-1):

    feature_l --MGAMaskHead_l--> logits_l --[feature_l, logits_l]--> MaskCBAM_l --> refined_l (-> Detect)        model/model.py:57-74
    seg_total = SegmentationLoss({p3,p4,p5: logits}, masks_multi)                                                  model/model.py:196-202
    total     = e^{-s_det} det_loss + s_det + e^{-s_seg} seg_total + s_seg                                         model/model.py:204-206
    backward of all of it; the gradient of feature_l = what MaskCBAM sends back + what its mask head sends back (the feature
    feeds both), accumulated in the head's GEMM epilogue instead of a feature-sized add.

Everything between the neck's P3/P4/P5 features and the Detect head's inputs -- i.e. every MGA-specific layer of the reference model
plus its loss terms -- runs as C-ABI calls on pre-allocated buffers: 3 + 2 + 2 launches forward, 1 + 2 + 5 backward for
all three levels together, recorded into one hipGraph.  The backbone / neck / Detect / detection loss are out of scope:
their contribution enters as given tensors -- `gy_l` (dL/d refined_l, what Detect's backward would deliver) and `det_loss`
(the criterion's 3-vector).

Buffers (per level l): x (feature), logits (= MaskCBAM's mask input, fp32), y, gy, gx, targets; one flat fp32 gradient bucket for
every parameter of the slice (three MaskCBAM blocks, three mask heads, the two Kendall log-variances): what DDP all-reduces."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import torch

from . import _lib
from .functional import BlockConfig, _head_params
from .plan import PyramidPlan

HEAD_PARAM_NAMES = ("proj.0.weight", "proj.1.weight", "proj.1.bias", "head.weight", "head.bias")


class SlicePlan:
    def __init__(self, shapes: Sequence[Tuple[int, int, int, int]], hidden: Sequence[int], cbam_params, cbam_cfgs: Sequence[BlockConfig],
                 head_states: Sequence[dict], target_hw: Sequence[Tuple[int, int]] = None, scale_weights=(1.0, 1.0, 1.0),
                 bn_eps: float = 1e-3, bn_momentum: float = 0.03, device="cuda", training: bool = True, dtype: torch.dtype = torch.float32):
        """shapes: (B,C,H,W) of the P3/P4/P5 features; hidden: mask-head widths; cbam_params: per level (w1,b1,w2,b2,wsa,beta);
        head_states: per level a MGAMaskHead state_dict; target_hw: resolution of the segmentation targets (default: feature size);
        bn_eps / bn_momentum: what Ultralytics' initialize_weights gives every BatchNorm2d (U/utils/torch_utils.py:570-572);
        dtype: element type of the FEATURES and their gradients (x, y, gy, gx: fp32 | fp16 | bf16).  The mask logits, the targets, every
        statistic and every parameter gradient stay fp32: the heads emit fp32 logits directly (MGAHEAD_LOGITS_F32), which is what
        MaskCBAM's mask input and the loss take -- no conversion pass anywhere in the slice."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        dev = self.device
        self.n = len(shapes)
        self.shapes, self.hidden = list(shapes), list(hidden)
        f32 = torch.float32
        # ---- one flat gradient bucket: [cbam grads of all levels][head grads of all levels][log_vars] ---------------------------------
        n_cbam = sum(p.numel() for ps in cbam_params for p in ps)
        self.head_params: List[List[torch.Tensor]] = []
        self.head_buffers: List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = []
        for sd in head_states:
            self.head_params.append([sd[k].detach().to(dev, f32).contiguous().clone() for k in HEAD_PARAM_NAMES])
            self.head_buffers.append((sd["proj.1.running_mean"].detach().to(dev, f32).clone(), sd["proj.1.running_var"].detach().to(dev, f32).clone(),
                                      sd["proj.1.num_batches_tracked"].detach().to(dev).clone()))
        n_head = sum(p.numel() for ps in self.head_params for p in ps)
        self.grad_bucket = torch.zeros(n_cbam + n_head + 2, dtype=f32, device=dev)
        self.dtype = dtype
        dcode = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[dtype]
        self.cbam = PyramidPlan(shapes, cbam_params, cbam_cfgs, dtype=dtype, device=dev, with_mask=True, want_gmask=True,
                                grad_bucket=self.grad_bucket[:n_cbam])
        self.x, self.logits, self.y, self.gy, self.gx = self.cbam.x, self.cbam.mask, self.cbam.y, self.cbam.gy, self.cbam.gx
        off = n_cbam
        self.head_grads: List[List[torch.Tensor]] = []
        for ps in self.head_params:
            views = []
            for p in ps:
                views.append(self.grad_bucket[off:off + p.numel()].view(p.shape))
                off += p.numel()
            self.head_grads.append(views)
        self.g_log_vars = self.grad_bucket[off:off + 2]
        # ---- loss side --------------------------------------------------------------------------------------------------------------
        self.log_vars = torch.zeros(2, dtype=f32, device=dev)          # mtl_log_vars (model.py:119-121)
        self.det_loss = torch.zeros(3, dtype=f32, device=dev)          # the detection criterion's (box, cls, dfl) vector: given
        self.total = torch.zeros(3, dtype=f32, device=dev)
        self.g_total = torch.ones(3, dtype=f32, device=dev)            # trainer: loss.sum().backward()
        self.g_det = torch.zeros(3, dtype=f32, device=dev)
        self.g_seg = torch.zeros((), dtype=f32, device=dev)
        target_hw = list(target_hw) if target_hw is not None else [(H, W) for _, _, H, W in shapes]
        self.targets = [torch.zeros(B, 1, th, tw, dtype=f32, device=dev) for (B, _, _, _), (th, tw) in zip(shapes, target_hw)]
        self.seg_glogits = [torch.zeros(B, 1, H, W, dtype=f32, device=dev) for B, _, H, W in shapes]
        self.seg_out = torch.zeros(1 + 3 * self.n, dtype=f32, device=dev)
        self._seg = (_lib.SegLevel * self.n)()
        for l, ((B, Cc, H, W), (th, tw)) in enumerate(zip(shapes, target_hw)):
            S = self._seg[l]
            S.logits, S.target, S.glogits = self.logits[l].data_ptr(), self.targets[l].data_ptr(), self.seg_glogits[l].data_ptr()
            S.B, S.H, S.W, S.Ht, S.Wt = B, H, W, th, tw
            S.dtype, S.scale_weight, S.resize = _lib.F32, float(scale_weights[l]), _lib.SEG_NEAREST
        self._seg_cfg = _lib.SegCfg(1.0, 1.0, 1.0, 1.0, 0, 0.5, 0.6, 0.5)      # SegLossConfig defaults (losses/segmentation.py:9-21)
        self.seg_ws = torch.zeros(self.lib.mgaseg_ws_bytes(self._seg, self.n), dtype=torch.uint8, device=dev)
        # ---- mask heads -------------------------------------------------------------------------------------------------------------
        self._hf, self._hb = (_lib.HeadFwdLevel * self.n)(), (_lib.HeadBwdLevel * self.n)()
        self.head_ctx, self.head_scratch = [], []
        for l, (B, Cc, H, W) in enumerate(shapes):
            hid = self.hidden[l]
            w1, gamma, beta, wh, bh = self.head_params[l]
            rm, rv, nbt = self.head_buffers[l]
            self.head_ctx.append(torch.zeros(self.lib.mgahead_ctx_bytes(B, Cc, H, W, hid), dtype=torch.uint8, device=dev))
            self.head_scratch.append(torch.zeros(self.lib.mgahead_bwd_scratch_bytes(B, Cc, H, W, hid), dtype=torch.uint8, device=dev))
            P = _head_params(w1, gamma, beta, rm, rv, nbt, wh, bh, hid, bn_eps, bn_momentum, training)
            F, Bw = self._hf[l], self._hb[l]
            F.x, F.logits, F.ctx, F.p = self.x[l].data_ptr(), self.logits[l].data_ptr(), self.head_ctx[l].data_ptr(), P
            F.ctx_bytes = Bw.ctx_bytes = self.head_ctx[l].numel()
            Bw.scratch_bytes = self.head_scratch[l].numel()
            F.B, F.C, F.H, F.W, F.dtype, F.flags = B, Cc, H, W, dcode, _lib.HEAD_LOGITS_F32
            gw1, gg, gb, gwh, gbh = self.head_grads[l]
            # dL/dlogits = the loss's part (seg_glogits) + MaskCBAM's dL/dmask (g_logits2): summed while the head's backward loads them
            Bw.x, Bw.g_logits, Bw.g_logits2, Bw.ctx, Bw.scratch, Bw.gx = (self.x[l].data_ptr(), self.seg_glogits[l].data_ptr(),
                                                                          self.cbam.gmask[l].data_ptr(), self.head_ctx[l].data_ptr(),
                                                                          self.head_scratch[l].data_ptr(), self.gx[l].data_ptr())
            Bw.gw1, Bw.gbn_weight, Bw.gbn_bias, Bw.gwh, Bw.gbh = gw1.data_ptr(), gg.data_ptr(), gb.data_ptr(), gwh.data_ptr(), gbh.data_ptr()
            Bw.p = _head_params(w1, gamma, beta, rm, rv, None, wh, bh, hid, bn_eps, bn_momentum, training)
            Bw.B, Bw.C, Bw.H, Bw.W, Bw.dtype, Bw.flags = B, Cc, H, W, dcode, _lib.HEAD_BWD_ACCUM_GX | _lib.HEAD_LOGITS_F32

    # ------------------------------------------------------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def forward(self):
        st = self._stream()
        _lib.check(self.lib.mgahead_forward(self._hf, self.n, st), "mgahead_forward")                 # features -> mask logits
        self.cbam.forward()                                                                            # [feature, logits] -> refined
        # logits, targets -> seg_total (+ log entries) -> Kendall total, the combine riding in the loss's last launch
        _lib.check(self.lib.mgaseg_kendall_forward(self._seg, self.n, C.byref(self._seg_cfg), self.seg_ws.data_ptr(), self.seg_ws.numel(), self.seg_out.data_ptr(),
                                                   self.det_loss.data_ptr(), 3, self.log_vars.data_ptr(), self.total.data_ptr(), st),
                   "mgaseg_kendall_forward")

    def backward(self):
        st = self._stream()
        _lib.check(self.lib.mgaseg_kendall_backward(self._seg, self.n, C.byref(self._seg_cfg), self.seg_ws.data_ptr(), self.seg_ws.numel(), self.seg_out.data_ptr(),
                                                    self.det_loss.data_ptr(), 3, self.log_vars.data_ptr(), self.g_total.data_ptr(),
                                                    self.g_det.data_ptr(), self.g_seg.data_ptr(), self.g_log_vars.data_ptr(), st),
                   "mgaseg_kendall_backward")                                                          # -> seg_glogits, g_det, g_log_vars
        self.cbam.backward()                                                                           # gy -> gx (MaskCBAM's part), dL/dmask (its part of dL/dlogits)
        _lib.check(self.lib.mgahead_backward(self._hb, self.n, st), "mgahead_backward")                # gx += head's part; head parameter gradients

    def step(self):
        self.forward()
        self.backward()

    capture = PyramidPlan.capture

    def check_handoff(self):
        self.cbam.check_handoff()

    def images(self) -> int:
        return self.shapes[0][0]

    def launches(self) -> dict:
        return dict(forward="3 (heads) + 2 (MaskCBAM) + 2 (seg loss + Kendall)",
                    backward="1 (seg loss + Kendall) + 2 (MaskCBAM) + 5 (heads)")
