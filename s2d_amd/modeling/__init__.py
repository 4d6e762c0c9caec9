from .backbone import ResNet50, build_resnet_backbone
from .pixel_decoder import MSDeformAttn, MSDeformAttnPixelDecoder
from .video_decoder import MaskOutputs, VideoMultiScaleMaskedTransformerDecoder
from .criterion import TargetSet, VideoHungarianMatcher, VideoSetCriterion
from .meta_arch import (KDVideoMaskFormer, MaskFormerHead, VideoMaskFormer, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY,
                        build_kd_model, set_amp_compute)
