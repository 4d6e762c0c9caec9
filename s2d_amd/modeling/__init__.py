from .backbone import ResNet50, build_resnet_backbone
from .pixel_decoder import MSDeformAttn, MSDeformAttnPixelDecoder
from .video_decoder import MaskOutputs, VideoMultiScaleMaskedTransformerDecoder
