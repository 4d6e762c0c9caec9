"""Clip frame selection of the dataset mapper (model_training/mask2former_video/data_video/dataset_mapper.py:223-289).
Host logic on a handful of integers; it consumes Python's `random` and `numpy.random` in the reference's order, so the same
seeds select the same frames (golden: tests/golden/sampling.json, produced by the reference's own methods)."""
import random

import numpy as np


def _sparse_window(video_length, num, frange, shuffle, clamp):
    ref = random.randrange(video_length)
    lo, hi = max(0, ref - frange), min(video_length, ref + frange + 1)
    others = list(range(lo, ref)) + list(range(ref + 1, hi))
    if clamp:                                           # :264-271: never more frames than there are, without replacement
        picked = np.random.choice(np.array(others), min(num - 1, len(others)), replace=False)
    else:                                               # :282-285: with replacement
        picked = np.random.choice(np.array(others), num - 1)
    sel = sorted(picked.tolist() + [ref])
    if shuffle:
        random.shuffle(sel)
    return sel


def dense_frame_selection(video_annos, video_length, sampling_frame_num, sampling_frame_range=5, sampling_frame_shuffle=False):
    """:223-274 -- a window of `sampling_frame_num` consecutive frames in which some instance is annotated throughout (one of
    all such windows, uniformly: instances in first-seen order, windows by start frame); if there is none, the sparse rule."""
    tracks = {}
    for t, annos in enumerate(video_annos):
        for a in annos:
            tracks.setdefault(a["id"], []).append(t)
    n = sampling_frame_num
    windows = []
    for frames in tracks.values():
        for i in range(len(frames) - n + 1):
            if frames[i + n - 1] - frames[i] == n - 1 and all(frames[i + j + 1] == frames[i + j] + 1 for j in range(n - 1)):
                windows.append(list(range(frames[i], frames[i] + n)))
    if windows:
        return random.choice(windows)
    return _sparse_window(video_length, n, sampling_frame_range, sampling_frame_shuffle, clamp=True)


def random_frame_selection(video_length, sampling_frame_num, sampling_frame_range=5, sampling_frame_shuffle=False):
    """:276-289 -- a reference frame and `sampling_frame_num - 1` of its neighbours within +-range (with replacement)"""
    return _sparse_window(video_length, sampling_frame_num, sampling_frame_range, sampling_frame_shuffle, clamp=False)
