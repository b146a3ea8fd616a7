#!/usr/bin/env python3
"""PCIe-inclusive rates of the step feeding the path, GEN1 B=5 T=32, 5 % occupancy.

  (a) reference-style feed: dense fp32 frames [T,B,2,H,W] built on the host, pinned H2D copy, NCHW -> NHWC pass
  (b) EventBatcher: raw events (4 x int32 per event) pinned H2D on a copy stream + HIP scatter into channels-last frames
and the training throughput with (b) inside the loop (the feed of step n+1 overlaps step n)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snn_for_object_detection_amd as S  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402

T, B, H, W, step_us = 32, 5, 240, 304, 1000
rng = np.random.default_rng(0)
samples = []
n_events = 0
for b in range(B):
    n = int(0.05 * T * 2 * H * W)
    t = rng.integers(0, T * step_us, n).astype(np.int32)
    x = rng.integers(0, W, n).astype(np.int32)
    y = rng.integers(0, H, n).astype(np.int32)
    p = rng.integers(0, 2, n).astype(np.int32)
    samples.append(tuple(torch.from_numpy(v).pin_memory() for v in (t, x, y, p)) + (0,))
    n_events += n
labels = [torch.tensor([[0, 0.2, 0.2, 0.5, 0.6], [1, 0.5, 0.4, 0.9, 0.8]])] * B
batcher = S.EventBatcher(T, H, W, step_us)
dense_host = batcher(samples).cpu().contiguous().pin_memory()   # what the reference's DataLoader would hand over


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def feed_dense():
    return S.functional.to_channels_last(dense_host.cuda(non_blocking=True))


def feed_events():
    return batcher(samples)


print(f"events per batch {n_events} ({16 * n_events / 1e6:.1f} MB as 4 x int32) vs dense {dense_host.numel() * 4 / 1e6:.1f} MB")
print(f"(a) dense H2D + layout pass : {timeit(feed_dense):6.2f} ms per batch")
print(f"(b) events H2D + scatter    : {timeit(feed_events):6.2f} ms per batch")

torch.manual_seed(2)
model = S.TinyYolo(num_classes=2, time_window=0).cuda().train()
trainer = FlatTrainer(model)
lab = torch.stack(labels).cuda()


def train_step(X):
    trainer.zero_grad()
    loss = model.training_step((X, lab))
    loss.backward()
    trainer.step()


X = feed_events()
resident = timeit(lambda: train_step(X), n=10)


def fed_step():
    global X
    train_step(X)          # queue step n ...
    X = feed_events()      # ... and feed step n+1 while it runs


fed = timeit(fed_step, n=10)
print(f"training step, batch resident in HBM : {resident:6.2f} ms -> {B * T / resident * 1e3:7.0f} event-frames/s")
print(f"training step, events fed every step : {fed:6.2f} ms -> {B * T / fed * 1e3:7.0f} event-frames/s (PCIe inclusive)")
