"""mga_yolo_amd -- MI355X-native (gfx950) implementation of ONE hot path of MGA-YOLO: the mask-guided CBAM block
(reference mga_yolo/nn/modules/masked_cbam.py) forward + backward, behind the reference's own module interface.

    from mga_yolo_amd import MaskCBAM, install
    install()                      # the reference's parse_model now builds this class for "MaskCBAM" YAML layers

Layout: ``csrc/`` hand-written HIP kernels + the C ABI of ``include/mgacbam.h``; ``_lib`` ctypes binding (no fallback);
``functional`` autograd entry points; ``module`` the nn.Module mirror; ``dp`` data-parallel gradient exchange (RCCL).
"""
from .functional import BlockConfig, EcaConfig, HandoffTimeout, handoff_report, mask_cbam, mask_cbam_pyramid, mask_eca, mask_eca_pyramid, mask_head, mask_head_pyramid, prob_mask_gate, resize_nearest  # noqa: F401
from .install import install, uninstall  # noqa: F401
from .module import MGAMaskHead, MaskCBAM, MaskECA, ProbMaskGater  # noqa: F401
from .segloss import SegLossConfig, SegmentationLoss, kendall_combine  # noqa: F401

__all__ = ["MaskCBAM", "MaskECA", "MGAMaskHead", "mask_head", "mask_head_pyramid", "ProbMaskGater", "BlockConfig", "EcaConfig", "mask_cbam", "mask_cbam_pyramid", "mask_eca",
           "mask_eca_pyramid", "prob_mask_gate", "resize_nearest", "install", "uninstall", "SegLossConfig", "SegmentationLoss", "kendall_combine", "HandoffTimeout", "handoff_report"]
__version__ = "0.1.0"
