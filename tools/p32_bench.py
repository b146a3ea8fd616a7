#!/usr/bin/env python3
"""Time snn_conv3x3_halo on the 32 -> 32 layer of the full-resolution stage (120x152, N = 160): forward (fp16 x 3, with
statistics partials) and data gradient (bf16 x 3, two fused addends), sustained, and both without their epilogue extras
(what the statistics / the two addends cost).  Round 4 used it to compare a persistent-block form of the kernel (DESIGN.md,
"A persistent form of the 32-channel rectangle kernel, built and not kept")."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

N, H, W, C = 160, 120, 152, 32
B = 5
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(N, H, W, C, device=dev)
w = torch.randn(C, 3, 3, C, device=dev) / (9 * C) ** 0.5
a1, a2 = torch.randn(N, H, W, C, device=dev), torch.randn(N, H, W, C, device=dev)
y = torch.empty(N, H, W, C, device=dev)


def image(prec):
    img = torch.empty(9 * C * C, device=dev)
    table = torch.tensor([[0, 0, C, C]], dtype=torch.int64, device=dev)
    _hip.call("snn_weight_frag_image_batched", w.data_ptr(), img.data_ptr(), table.data_ptr(), 1, 9 * 128, 0, prec, st)
    return img


img_f, img_b = image(_hip.PREC_FP16X3), image(_hip.PREC_BF16X3)
n_part = _hip.query("snn_conv2d_fwd_bn_partial_size", N, B, H, W, C)
partial = torch.empty((n_part,), device=dev, dtype=torch.float64)
layout = (ctypes.c_int * 2)()


def fwd():
    _hip.call("snn_conv3x3_halo", x.data_ptr(), C, img_f.data_ptr(), y.data_ptr(), C, N, H, W, C, C, None, 0, None, 0,
              partial.data_ptr(), B, layout, _hip.PREC_FP16X3, st)


def dgrad():
    _hip.call("snn_conv3x3_halo", x.data_ptr(), C, img_b.data_ptr(), y.data_ptr(), C, N, H, W, C, C, a1.data_ptr(), C,
              a2.data_ptr(), C, None, 0, None, _hip.PREC_BF16X3, st)


def fwd_plain():
    _hip.call("snn_conv3x3_halo", x.data_ptr(), C, img_f.data_ptr(), y.data_ptr(), C, N, H, W, C, C, None, 0, None, 0,
              None, 0, None, _hip.PREC_FP16X3, st)


def dgrad_plain():
    _hip.call("snn_conv3x3_halo", x.data_ptr(), C, img_b.data_ptr(), y.data_ptr(), C, N, H, W, C, C, None, 0, None, 0,
              None, 0, None, _hip.PREC_BF16X3, st)


for name, fn in (("forward + statistics", fwd), ("data gradient + 2 addends", dgrad), ("forward, no statistics", fwd_plain),
                 ("data gradient, no addend", dgrad_plain)):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10.0
    print(f"{name:28s} {us:8.1f} us   {2.0 * N * H * W * C * 9 * C / us / 1e6:6.1f} TFLOP/s")
