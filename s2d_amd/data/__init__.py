"""Data-side step before the hot path: clip frame selection, clip augmentation and video copy-paste, with
the frames and instance masks resident on the GPU (csrc/augment.hip) instead of numpy / PIL passes in dataloader workers."""
from .sampling import dense_frame_selection, random_frame_selection  # noqa: F401
from .augment import ClipAugmentation, augment_clip  # noqa: F401
from .copy_paste import copy_and_paste, copy_and_paste_clip, propagate_sparse_masks  # noqa: F401
from .assemble import assemble_clip_instances, clip_id_slots  # noqa: F401
