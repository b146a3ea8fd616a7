#!/usr/bin/env python3
"""Which torch (non-library) ops run in a step, with input shapes and source lines (torch.profiler)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import TinyYolo  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
model = TinyYolo(num_classes=2, time_window=0).to(dev).train()
trainer = FlatTrainer(model)
T, B = 32, 5
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=50,
                                                         max_shapes_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=40,
                                                  max_src_column_width=90))
