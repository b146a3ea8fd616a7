#!/usr/bin/env python3
"""Forward conv arithmetic modes vs fp64: relative L2 / max errors for well- and badly-scaled operands."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import functional as HF  # noqa: E402

torch.manual_seed(0)
cases = [("unit activations, Kaiming weights", 1.0, None), ("activations x 300", 300.0, None),
         ("activations x 1e-3", 1e-3, None), ("weights x 1e-3", 1.0, 1e-3), ("weights x 30", 1.0, 30.0),
         ("spikes {0,1,2}", "spikes", None)]
for name, xs, ws in cases:
    Cin, Cout, k, H, W = 128, 128, 3, 30, 38
    x = (torch.randint(0, 3, (2, 2, Cin, H, W)).float() if xs == "spikes" else torch.randn(2, 2, Cin, H, W) * xs)
    w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5 * (ws or 1.0)
    ref = F.conv2d(x.double().flatten(0, 1), w.double(), padding=1)
    row = []
    for mode in ("fp32", "bf16x6", "fp16x3"):
        HF.set_forward_precision(mode)
        y = HF.conv2d(x.cuda(), w.cuda(), stride=1, padding=1).flatten(0, 1).double().cpu()
        l2 = float((y - ref).norm() / ref.norm())
        mx = float((y - ref).abs().max() / ref.abs().max())
        row.append(f"{mode}: L2 {l2:.2e} max {mx:.2e}")
    print(f"{name:36s} " + " | ".join(row))
HF.set_forward_precision("bf16x6")
