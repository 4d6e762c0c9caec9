from .propagate import (color_masks_to_ids, compute_point_mask_intersection, extract_mask_matches, local_correlation, pred_tracks_to_binary_masks,
                        visibility_curve, IdMap)
from .grouping import temporal_groups, visibility_windows
