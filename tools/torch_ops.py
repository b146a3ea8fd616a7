#!/usr/bin/env python3
"""Which torch (ATen) device kernels are left in one GEN1 training step, and where they come from.
usage: torch_ops.py   (prints op -> count, device time and the innermost repo frames that issued it)"""
import os
import sys
from collections import defaultdict

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snn_for_object_detection_amd as S  # noqa: E402
from bench import synthetic_batch  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(2)
X, labels = synthetic_batch(32, 5, 240, 304, 2, dev, seed=0, p=0.05)
model = S.TinyYolo(num_classes=2, time_window=0).to(dev).train()
tr = FlatTrainer(model, lr=1e-3)


def step():
    tr.zero_grad()
    loss = model.training_step((X, labels))
    loss.backward()
    tr.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children and any(
            c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue
    frames = [f for f in (ev.stack or []) if root in f and "tools/torch_ops.py" not in f][:2]
    key = (ev.name, " <- ".join(f.replace(root + "/", "") for f in frames) or "(autograd engine / torch internals)")
    agg[key][0] += 1
    agg[key][1] += ev.device_time_total
tot = sum(v[1] for v in agg.values())
print(f"leaf ATen ops with device time in one step: {sum(v[0] for v in agg.values())} launches, {tot:.0f} us")
for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {name:28s} {us:8.1f} us   {where}")
