#!/usr/bin/env python3
"""In-kernel clock of the implicit-GEMM convolution under its own load (MI355X lowers its clock under MFMA load: the
nominal peaks are quoted at 2.4 GHz).  Needs the -DSNN_CLOCK build: python -m snn_for_object_detection_amd._build --clock,
SNN_HIP_LIB=build/libsnn_hip_clock.so.
usage: clock_conv.py fwd|dgrad Cin Cout k s H W [frames] [seconds]
Every block stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its begin and end; after >= `seconds` of
back-to-back launches on random data the clock is delta(cycles) / delta(10 ns ticks) x 100 MHz, median over blocks."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

op, Cin, Cout, k, s, H, W = sys.argv[1], *map(int, sys.argv[2:8])
N = int(sys.argv[8]) if len(sys.argv) > 8 else 160
seconds = float(sys.argv[9]) if len(sys.argv) > 9 else 2.5
lib = _hip.load()
dev, st, pad = torch.device("cuda"), torch.cuda.current_stream().cuda_stream, k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
wt = torch.randn(Cin, k, k, Cout, device=dev) * 0.05
y = torch.empty(N, Ho, Wo, Cout, device=dev)
dy = torch.randn(N, Ho, Wo, Cout, device=dev)
dx = torch.empty_like(x)


def launch():
    if op == "fwd":
        _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k, k, s,
                  pad, None, 0, None, 0, None, 4, st)
    else:
        _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout, k,
                  k, s, pad, None, 0, None, 0, 1, st)


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
t0, n = time.time(), 0
while time.time() - t0 < seconds:
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    n += 200
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
us_hot = e0.elapsed_time(e1) * 1e3 / 20
buf = np.zeros(2048 * 4, dtype=np.uint64)
lib.snn_debug_stamps2.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.snn_debug_stamps2(buf.ctypes.data, 2048 * 4) == 0
b = buf.reshape(2048, 4).astype(np.float64)
b = b[(b[:, 1] > b[:, 0]) & (b[:, 3] > b[:, 2])]
ghz = (b[:, 1] - b[:, 0]) / (b[:, 3] - b[:, 2]) * 0.1
flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
print(f"{op} {Cin}->{Cout} k{k} s{s} {H}x{W} N={N}: {us:.1f} us cold, {us_hot:.1f} us after {n} launches "
      f"({flops / us_hot / 1e6:.1f} TFLOP/s); in-kernel clock median {np.median(ghz):.3f} GHz "
      f"(p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}) over {len(ghz)} blocks; "
      f"block life {np.median(b[:, 1] - b[:, 0]):.0f} cycles")
