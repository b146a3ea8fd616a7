#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU's share: the deep 12 x {Conv(64,3), Norm, LIF} backbone at T=128, B=16/8 = 2 per
GPU, forward + BPTT backward + flat-gradient Adamax.   usage: deep12_probe.py [H W B T]   (default 240 304 2 128)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import BlockGen, Conv, LIF, Norm  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402

H, W, B, T = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (240, 304, 2, 128)
dev = torch.device("cuda")
torch.manual_seed(5)
cfg = []
for _ in range(12):
    cfg += [Conv(64, 3), Norm(), LIF()]
net = BlockGen(2, cfg)
for m in net.modules():
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
net = net.to(dev).train()
tr = FlatTrainer(net)
X = (torch.rand(T, B, 2, H, W, device=dev) < 0.3).float()
probe = torch.randn(B, 64, H, W, device=dev)


def step():
    tr.zero_grad()
    out, _ = net(X)
    loss = (out[-1] * probe).mean()   # read-out at the last step, gradient flows back through all T steps
    loss.backward()
    tr.step()
    return loss, out


for _ in range(2):
    step()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    loss, out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = 3 * 2.0 * T * B * H * W * 64 * 9 * (2 + 11 * 64)
print(f"deep12 {W}x{H} B={B} T={T}: peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, reserved "
      f"{torch.cuda.max_memory_reserved() / 2**30:.1f} GiB, {1e3 * dt:.1f} ms/step, {B * T / dt:.0f} event-frames/s, "
      f"conv {flops / dt / 1e12:.0f} TFLOP/s, last-layer rate {float(out.detach().mean()):.4f}, "
      f"loss {float(loss.detach()):.5f}")
