#!/usr/bin/env python3
"""Time snn_conv2d_wgrad (kernel + ordered slab reduce) on the 3x3 layer shapes of TinyYolo GEN1 B=5 T=32 (and the
deep-12 / 1 Mpx shapes with --all).  With a -DSNN_TUNING library (SNN_HIP_LIB=build/libsnn_hip_tuning.so) and
SNN_WGRAD_NO_HALO=1 the same shapes run on the implicit-GEMM kernel for comparison."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, stride, launches per step
    (160, 120, 152, 32, 32, 1, 2), (160, 60, 76, 64, 64, 1, 3), (160, 30, 38, 128, 128, 1, 4),
    (160, 15, 19, 128, 128, 1, 3), (160, 8, 10, 128, 128, 1, 2), (160, 120, 152, 64, 128, 2, 1),
    (160, 60, 76, 128, 256, 2, 1), (160, 30, 38, 256, 256, 2, 1), (160, 15, 19, 256, 256, 2, 1),
]
if "--all" in sys.argv:
    SHAPES += [(256, 240, 304, 64, 64, 1, 11), (64, 360, 640, 32, 32, 1, 2), (64, 180, 320, 64, 64, 1, 3)]
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
total = 0.0
for N, H, W, Cin, Cout, s, mult in SHAPES:
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev)
    dy = torch.randn(N, Ho, Wo, Cout, device=dev) * 1e-3
    dw = torch.empty(Cout, 3, 3, Cin, device=dev)
    splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, 3, 3, s, 1, _hip.PREC_BF16X3)
    ws = torch.empty(splitk, dw.numel(), device=dev)

    def call():
        _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Ho, Wo, Cout,
                  3, 3, s, 1, 0, ws.data_ptr(), splitk, _hip.PREC_BF16X3, st)
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100.0
    flops = 2.0 * N * Ho * Wo * Cout * 9 * Cin
    total += us * mult
    print(f"{Cin:4d}->{Cout:4d} s{s} {H:3d}x{W:3d} N={N}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  slabs {splitk}")
print(f"sum over the launches of a step: {total / 1e3:.2f} ms")
