"""sfvos_amd -- MI355X-native SlowFastLayers hot path (hand-written HIP behind a C ABI).

Public surface mirrors the reference's `helpers/model.py` for this path:
    SlowFastLayers(input_size, device, slow_pathway_size, fast_pathway_size)
plus the pieces the reference's train.py builds around it (SGD step, gradient averaging for
one-process-per-GPU data parallelism)."""
from .plan import SlowFastPlan, calc_fuse_kernel_size, calc_kernel_sizes, davis_pyramid  # noqa: F401
from .module import PackedClip, SlowFastLayers, union_mask  # noqa: F401
from .optim import FusedSGD  # noqa: F401
from .losses import MSEProxyLoss  # noqa: F401
from .stream import SlowFastStream  # noqa: F401
from .mask_head import (MaskBranch, MaskRCNNHeads, MaskRCNNPredictor, maskrcnn_inference,  # noqa: F401
                        maskrcnn_loss, paste_masks_in_image)
from .parallel import GradBucket, init_distributed  # noqa: F401
from .roi_align import MultiScaleRoIAlign, roi_align  # noqa: F401
from .graph import GraphedStep  # noqa: F401

__all__ = ['SlowFastLayers', 'PackedClip', 'SlowFastPlan', 'FusedSGD', 'GradBucket', 'init_distributed',
           'MSEProxyLoss', 'calc_kernel_sizes', 'calc_fuse_kernel_size', 'davis_pyramid', 'union_mask', 'SlowFastStream', 'MaskBranch', 'MaskRCNNHeads', 'MaskRCNNPredictor',
           'maskrcnn_inference', 'maskrcnn_loss', 'paste_masks_in_image', 'MultiScaleRoIAlign', 'roi_align', 'GraphedStep']
