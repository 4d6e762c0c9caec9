from .propagate import (color_masks_to_ids, compute_point_mask_intersection, extract_mask_matches, local_correlation, point_id_counts,
                        pred_tracks_to_binary_masks, visibility_curve, IdMap)
from .grouping import temporal_groups, visibility_windows
from . import formats
from .formats import (merge_ytvis_jsons, save_segmentation_masks, save_temporal_group_masks, select_masks,
                      write_annotation_for_video)
