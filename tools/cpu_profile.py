#!/usr/bin/env python3
"""Host-side cost of one training step: cProfile over launch-bound steps (T=2), top functions by own time."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import TinyYolo  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
model = TinyYolo(num_classes=2, time_window=0).to(dev).train()
trainer = FlatTrainer(model)
T, B = 2, 5
X = (torch.rand(T, B, 2, 240, 304, device=dev) < 0.05).float()
labels = torch.tensor([[[0, 0.2, 0.2, 0.5, 0.6], [1, 0.5, 0.4, 0.9, 0.8]]] * B, device=dev)


def step():
    trainer.zero_grad()
    loss = model.training_step((X, labels), 0)
    loss.backward()
    trainer.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
