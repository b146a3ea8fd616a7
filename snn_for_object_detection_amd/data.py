"""The step FEEDING the path: raw events -> a device batch ``X[T, B, 2, H, W]``.

The reference voxelises on the CPU inside DataLoader workers (``utils/datasets.py:403-435``) and ships dense fp32
frames to the GPU: 93 MB per B=5, T=32 GEN1 batch, 1.9 GB for the 1Mpx batch.  Here the host hands over the raw
events (16 bytes each: 5 % occupancy = 19 MB for the GEN1 batch), the copy runs on a side stream out of pinned
memory, and a HIP scatter kernel (``snn_events_to_frames``) writes the binary frames directly in the channels-last
layout the first convolution reads, so the NCHW -> NHWC pass over the input disappears as well.

``EventBatcher`` mirrors the reference's sample / collate contract:
* per sample: events ``(t_us, x, y, p)`` with ``t >= t0`` are binned ``(t - t0) // time_step_us`` into ``num_steps``
  frames, ``x`` clipped to the frame, value 1 not a count (``datasets.py:415-433``);
* the batch stacks samples on dim 1 and pads label rows with -1 (``_stack_data``, ``datasets.py:127-135``).
"""
from typing import List, Optional, Sequence, Tuple

import torch

from . import _hip


class EventBatcher:
    def __init__(self, num_steps: int, height: int, width: int, time_step_us: int, device="cuda"):
        self.T, self.H, self.W, self.step = int(num_steps), int(height), int(width), int(time_step_us)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("EventBatcher: a HIP device is required (no CPU fallback)")
        # a stream that really runs beside the main stream (HIP streams share a few hardware queues: functional.concurrent_stream)
        from . import functional as _HF
        dev = self.device if self.device.index is not None else torch.device("cuda", torch.cuda.current_device())
        self._copy_stream = _HF.concurrent_stream(torch.cuda.current_stream(dev), avoid=tuple(_HF._SIDE_STREAMS.values()))

    def __call__(self, samples: Sequence[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, int]],
                 labels: Optional[Sequence[torch.Tensor]] = None):
        """``samples[b] = (t_us, x, y, p, t0_us)`` - 1-D host tensors (int64 / int32; pinned memory makes the copy
        asynchronous) and the time of the first frame.  Returns ``X[T, B, 2, H, W]`` (logical NCHW view of a
        channels-last buffer) and, when ``labels`` is given, ``labels[B, N, 5]`` padded with -1."""
        B = len(samples)
        T, H, W = self.T, self.H, self.W
        main = torch.cuda.current_stream(self.device)
        buf = torch.empty((T, B, H, W, 2), device=self.device, dtype=torch.float32)   # [T][B][H][W][C]
        self._copy_stream.wait_stream(main)  # the allocation (and earlier users of its memory) precede the scatter
        with torch.cuda.stream(self._copy_stream):
            slots, xs, ys, ps = [], [], [], []
            for b, (t_us, x, y, p, t0) in enumerate(samples):
                if int(t_us.numel()) == 0:
                    continue
                dev = [v.to(self.device, non_blocking=True) for v in (t_us, x, y, p)]
                t_bin = (dev[0].to(torch.int64) - int(t0)).div(self.step, rounding_mode="floor")
                # frame index inside the batch buffer = t * B + b; events before t0 or past the window are dropped
                ok = (t_bin >= 0) & (t_bin < T)
                slots.append(torch.where(ok, t_bin * B + b, torch.full_like(t_bin, -1)).to(torch.int32))
                xs.append(dev[1].to(torch.int32))
                ys.append(dev[2].to(torch.int32))
                ps.append(dev[3].to(torch.int32))
            if slots:
                ev = [torch.cat(v) for v in (slots, xs, ys, ps)]
                n = int(ev[0].numel())
            else:
                ev, n = [torch.zeros(1, device=self.device, dtype=torch.int32)] * 4, 0
            # one launch for the whole batch: zero-fills the buffer, then scatters every sample's events
            _hip.call("snn_events_to_frames", ev[0].data_ptr(), ev[1].data_ptr(), ev[2].data_ptr(), ev[3].data_ptr(), n,
                      buf.data_ptr(), T * B, H, W, self._copy_stream.cuda_stream)
            buf.record_stream(self._copy_stream)
        main.wait_stream(self._copy_stream)
        X = buf.permute(0, 1, 4, 2, 3)  # [T, B, 2, H, W], channels-last memory
        if labels is None:
            return X
        n_max = max(int(l.shape[0]) for l in labels) if labels else 0
        lab = torch.full((B, n_max, 5), -1.0, dtype=torch.float32)
        for b, l in enumerate(labels):
            lab[b, : l.shape[0]] = l
        return X, lab.to(self.device, non_blocking=True)
