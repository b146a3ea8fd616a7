#!/usr/bin/env python3
"""Overfit ONE synthetic GEN1 batch (B=5, T=32) for N steps with the flat-buffer trainer: the loss curve of the full
path (forward, last-step loss, BPTT, fused Adamax).   usage: train_demo.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import snn_for_object_detection_amd as S  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda")
torch.manual_seed(2)
model = S.TinyYolo(num_classes=2, time_window=0).to(dev).train()
trainer = FlatTrainer(model, lr=model.hparams.learning_rate)
X, labels = bench.synthetic_batch(32, 5, 240, 304, 2, dev, seed=0)
t0 = time.perf_counter()
for k in range(steps):
    trainer.zero_grad()
    loss = model.training_step((X, labels))
    loss.backward()
    trainer.step()
    if k % 20 == 0 or k == steps - 1:
        print(f"step {k:4d}  loss {float(loss.detach()):.5f}  ({time.perf_counter() - t0:.1f} s)", flush=True)
model.eval()
with torch.no_grad():
    anchors, cls, box = model(X)
print("finite:", bool(torch.isfinite(cls).all() and torch.isfinite(box).all()),
      " positive-class anchors:", int((cls.argmax(-1) > 0).sum()))
