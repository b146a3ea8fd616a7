#!/usr/bin/env python3
"""Detection decode (softmax probs -> NMS'd rows): device kernels vs the host torch path, GEN1 anchor count."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import box  # noqa: E402

torch.manual_seed(21)
A, K = 13545, 3
centers = torch.rand(A, 2)
wh = 0.02 + 0.1 * torch.rand(A, 2)
anchors = torch.cat([centers - wh / 2, centers + wh / 2], dim=1)
probs = torch.softmax(4 * torch.randn(1, A, K), dim=2)
offs = 0.3 * torch.randn(1, A, 4)
t0 = time.perf_counter()
ref = box.multibox_detection(probs.clone(), offs.clone(), anchors)
t_host = time.perf_counter() - t0
pd, od, ad = probs.cuda(), offs.cuda(), anchors.cuda()
for _ in range(3):
    box.multibox_detection(pd, od, ad)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    det = box.multibox_detection(pd, od, ad)
torch.cuda.synchronize()
t_dev = (time.perf_counter() - t0) / 20
print(f"A={A}: kept {int((ref[0, :, 0] >= 0).sum())}; host torch path {1e3 * t_host:.1f} ms; device path {1e3 * t_dev:.2f} ms "
      f"per frame (no host synchronisation)")
