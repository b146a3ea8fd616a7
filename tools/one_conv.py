#!/usr/bin/env python3
"""Run ONE conv entry point a few times (for rocprofv3 --pmc passes).  usage: one_conv.py fwd|dgrad|wgrad Cin Cout k s H W [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

op, Cin, Cout, k, s, H, W = sys.argv[1], *map(int, sys.argv[2:8])
N = int(sys.argv[8]) if len(sys.argv) > 8 else 160
_hip.load()
dev, st, pad = torch.device("cuda"), torch.cuda.current_stream().cuda_stream, k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
wt = torch.randn(Cin, k, k, Cout, device=dev) * 0.05
y = torch.empty(N, Ho, Wo, Cout, device=dev)
dy = torch.randn(N, Ho, Wo, Cout, device=dev)
dx, dw = torch.empty_like(x), torch.empty_like(w)
splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, 1)
ws = torch.empty(splitk * w.numel(), device=dev)
for _ in range(3):
    if op == "fwd":
        _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, None, 4, st)
    elif op == "dgrad":
        _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, 1, st)
    else:
        _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, 0, ws.data_ptr(), splitk, 1, st)
torch.cuda.synchronize()
