"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's event -> frame voxelisation and batch collate.

Follows ``utils/datasets.py`` of the reference (read as text; the module itself needs the absent
``prophesee_toolbox`` submodule, so it cannot be imported - PARITY UNPINNED by reference outputs, pinned by the
hand-checked vectors of ``tests/test_host_logic.py``):

* ``voxelize`` - ``STPropheseeDataset.parse_data`` (utils/datasets.py:403-435): events with ``t >= t0`` are binned
  ``(t - t0) // time_step_us``, ``x`` is clipped to ``[0, W-1]`` (1Mpx recordings contain events past the frame,
  :425-426) and ``features[bin, p, y, x] = 1`` (a flag, not a count; :428-433);
* ``stack_batch`` - ``_stack_data`` (utils/datasets.py:127-135): features stacked on dim 1 (``[T, B, 2, H, W]``),
  label rows padded with -1 to the longest sample.
"""
from typing import List, Sequence, Tuple

import numpy as np


def voxelize(t_us: np.ndarray, x: np.ndarray, y: np.ndarray, p: np.ndarray, t0_us: int, time_step_us: int,
             num_steps: int, height: int, width: int) -> np.ndarray:
    frames = np.zeros((num_steps, 2, height, width), dtype=np.float32)
    keep = t_us >= t0_us                                            # datasets.py:415
    t_us, x, y, p = t_us[keep], x[keep], y[keep], p[keep]
    if t_us.size == 0:
        return frames
    bins = (t_us - t0_us) // time_step_us                           # datasets.py:419
    inside = bins < num_steps                                        # load_delta_t bounds the window (datasets.py:412-414)
    bins, x, y, p = bins[inside], x[inside], y[inside], p[inside]
    x = np.clip(x, 0, width - 1)                                     # datasets.py:425-426
    frames[bins.astype(np.int64), p.astype(np.int64), y.astype(np.int64), x.astype(np.int64)] = 1   # :428-433
    return frames


def stack_batch(samples: Sequence[Tuple[np.ndarray, np.ndarray]]) -> Tuple[np.ndarray, np.ndarray]:
    feats = np.stack([s[0] for s in samples], axis=1)
    n = max(s[1].shape[0] for s in samples)
    labels = np.full((len(samples), n, samples[0][1].shape[1] if samples[0][1].ndim == 2 else 5), -1.0, dtype=np.float32)
    for b, (_, lab) in enumerate(samples):
        labels[b, : lab.shape[0]] = lab
    return feats, labels
