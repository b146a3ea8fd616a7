"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's event -> frame voxelisation and batch collate.

Follows ``utils/datasets.py`` of the reference.  PINNED on reference outputs since round 4: ``tests/golden/events.npz``
holds what the reference's own ``STPropheseeDataset.parse_data``, ``MTPropheseeDataset.parse_data`` and
``PropheseeDataModule._stack_data`` produced for seeded event streams (``tests/golden/make_golden.py::events_golden``
executes the method bodies unmodified in the build container; the recording reader they are handed is a fake object
with the three members they use), and ``tests/test_oracle_pins.py`` checks every function below against it bit for bit.

* ``voxelize`` - the scatter of ``STPropheseeDataset.parse_data`` (utils/datasets.py:403-435): events with ``t >= t0``
  are binned ``(t - t0) // time_step_us``, ``x`` is clipped to ``[0, W-1]`` (1Mpx recordings contain events past the
  frame, :425-426) and ``features[bin, p, y, x] = 1`` (a flag, not a count; :428-433);
* ``st_sample`` - the whole of that method (:378-435): label selection (first labelled step at or after
  ``start_step + num_steps``; boxes under the area threshold dropped), the window it reads from the recording, the
  rejection of sparse windows (``events // num_steps < events_threshold``);
* ``mt_sample`` - ``MTPropheseeDataset.parse_data`` (:311-344): ``bin = t // step - start_step``, NO clipping of x,
  labels of ``[start_step, start_step + num_steps)`` with their step made window-relative;
* ``stack_batch`` - ``_stack_data`` (:127-135): features stacked on dim 1 (``[T, B, 2, H, W]``), label rows padded with
  -1 to the longest sample.
"""
from typing import Optional, Sequence, Tuple

import numpy as np


def voxelize(t_us: np.ndarray, x: np.ndarray, y: np.ndarray, p: np.ndarray, t0_us: int, time_step_us: int,
             num_steps: int, height: int, width: int, clip_x: bool = True) -> np.ndarray:
    frames = np.zeros((num_steps, 2, height, width), dtype=np.float32)
    keep = t_us >= t0_us                                            # datasets.py:415
    t_us, x, y, p = t_us[keep], x[keep], y[keep], p[keep]
    if t_us.size == 0:
        return frames
    bins = (t_us - t0_us) // time_step_us                           # datasets.py:419
    inside = bins < num_steps                                        # load_delta_t bounds the window (datasets.py:412-414)
    bins, x, y, p = bins[inside], x[inside], y[inside], p[inside]
    if clip_x:
        x = np.clip(x, 0, width - 1)                                 # datasets.py:425-426
    frames[bins.astype(np.int64), p.astype(np.int64), y.astype(np.int64), x.astype(np.int64)] = 1   # :428-433
    return frames


def st_sample(gt_boxes: np.ndarray, t_us: np.ndarray, x: np.ndarray, y: np.ndarray, p: np.ndarray, clock_us: int,
              num_steps: int, time_shift: int, time_step_us: int, height: int, width: int, events_threshold: int = 4000,
              box_size_threshold: float = 0.01
              ) -> Tuple[Optional[Tuple[np.ndarray, np.ndarray]], bool, int, Optional[int]]:
    """``STPropheseeDataset.parse_data`` on a recording whose clock stands at ``clock_us``; ``gt_boxes`` rows are
    ``(step, class, x1, y1, x2, y2)``.  -> ``(sample or None, more, clock afterwards, t0 of the window or None)`` with
    ``sample = (features[T, 2, H, W], labels[n, 5])``."""
    start_step = clock_us // time_step_us                            # :386-387
    gt = gt_boxes[gt_boxes[:, 0] >= start_step + num_steps]          # :388
    if gt.size == 0:
        return None, False, clock_us, None
    labels = gt[gt[:, 0] == gt[0, 0]]                                # :391
    area = (labels[:, 4] - labels[:, 2]) * (labels[:, 5] - labels[:, 3])
    labels = labels[area > np.float32(box_size_threshold)]           # :394-397
    if labels.size == 0:
        return None, False, clock_us, None
    first_label_us = int(labels[0, 0]) * time_step_us                # :408
    t0 = first_label_us - time_step_us * (num_steps - time_shift)    # :409-411
    delta = first_label_us + time_step_us * time_shift - clock_us    # :412-414: the recording is read up to the window end
    window = (t_us >= clock_us) & (t_us < clock_us + delta)
    clock_after = clock_us + delta
    t_w, x_w, y_w, p_w = t_us[window], x[window], y[window], p[window]
    keep = t_w >= t0                                                 # :415
    if (int(keep.sum()) // num_steps) < events_threshold:            # :416-417
        return None, True, clock_after, t0
    feats = voxelize(t_w, x_w, y_w, p_w, t0, time_step_us, num_steps, height, width)
    return (feats, labels[:, 1:]), True, clock_after, t0


def mt_sample(gt_boxes: np.ndarray, t_us: np.ndarray, x: np.ndarray, y: np.ndarray, p: np.ndarray, clock_us: int,
              num_steps: int, time_step_us: int, height: int, width: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """``MTPropheseeDataset.parse_data``: -> ``(features[T, 2, H, W], labels[n, 6] with window-relative step, clock)``."""
    start_step = clock_us // time_step_us                            # :324
    duration = time_step_us * num_steps
    window = (t_us >= clock_us) & (t_us < clock_us + duration)       # :326
    t_w = t_us[window]
    if t_w.size == 0:                                                # :328-329
        return np.zeros((num_steps, 2, height, width), np.float32), gt_boxes[0:0], clock_us + duration
    # bin = t // step - start_step (:327) = (t - start_step * step) // step; no clipping in this method
    feats = voxelize(t_w, x[window], y[window], p[window], start_step * time_step_us, time_step_us, num_steps, height,
                     width, clip_x=False)
    sel = (gt_boxes[:, 0] >= start_step) & (gt_boxes[:, 0] < start_step + num_steps)   # :340
    labels = gt_boxes[sel].copy()
    labels[:, 0] -= start_step                                       # :341
    return feats, labels, clock_us + duration


def stack_batch(samples: Sequence[Tuple[np.ndarray, np.ndarray]]) -> Tuple[np.ndarray, np.ndarray]:
    feats = np.stack([s[0] for s in samples], axis=1)
    n = max(s[1].shape[0] for s in samples)
    labels = np.full((len(samples), n, samples[0][1].shape[1] if samples[0][1].ndim == 2 else 5), -1.0, dtype=np.float32)
    for b, (_, lab) in enumerate(samples):
        labels[b, : lab.shape[0]] = lab
    return feats, labels
