#!/usr/bin/env python3
"""Micro-benchmark of the conv entry points on the dominant TinyYolo GEN1 shapes.

    python tools/conv_bench.py [--frames 160] [--iters 20] [--only fwd|dgrad|wgrad]

Times each C-ABI call with HIP events on the launch stream and prints TFLOP/s (algorithmic FLOPs).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

SHAPES = [  # Cin, Cout, k, s, H, W (input size), label
    (128, 128, 3, 1, 30, 38, "c3(128) x4"),
    (128, 128, 3, 1, 15, 19, "c3(128) neck2 x3"),
    (128, 128, 3, 1, 8, 10, "c3(128) neck3 x2"),
    (64, 64, 3, 1, 60, 76, "c3(64) x3"),
    (32, 32, 3, 1, 120, 152, "c3(32) x2"),
    (64, 128, 3, 2, 120, 152, "down 64->128"),
    (128, 256, 3, 2, 60, 76, "down 128->256"),
    (256, 256, 3, 2, 30, 38, "down 256->256"),
    (768, 256, 1, 1, 30, 38, "c2f out 768->256"),
    (320, 128, 1, 1, 60, 76, "c2f out 320->128"),
    (128, 64, 1, 1, 120, 152, "c2f out 128->64"),
    (64, 64, 1, 1, 120, 152, "c2f in 64->64"),
    (2, 64, 3, 2, 240, 304, "first conv"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    _hip.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    N = args.frames
    print(f"{'shape':28s} {'op':6s} {'us':>9s} {'TFLOP/s':>8s}")
    for Cin, Cout, k, s, H, W, label in SHAPES:
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        x = torch.randn(N, H, W, Cin, device=dev)
        if Cin == 2:  # the event-frame layer sees sparse binary input
            x = (torch.rand(N, H, W, Cin, device=dev) < float(os.environ.get("EVENT_P", "0.05"))).float()
        w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
        wt = torch.empty(Cin, k, k, Cout, device=dev)
        y = torch.empty(N, Ho, Wo, Cout, device=dev)
        dy = torch.randn(N, Ho, Wo, Cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        _hip.call("snn_weight_transpose", w.data_ptr(), wt.data_ptr(), Cout, k, k, Cin, st)
        splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, 1)
        ws = torch.empty(splitk * w.numel(), device=dev)
        flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
        ops = {
            "fwd": lambda: _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W,
                                     Cin, Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, None, 4, st),
            "dgrad": lambda: _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N,
                                       H, W, Cin, Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, 1, st),
            "wgrad": lambda: _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N,
                                       H, W, Cin, Ho, Wo, Cout, k, k, s, pad, 0, ws.data_ptr(), splitk, 1, st),
        }
        if k == 3 and s == 1 and _hip.query("snn_conv3x3_halo_supported", N, H, W, Cin, Cout):
            def image(src, O, I, flip, prec):
                img = torch.empty(9 * O * I, device=dev)
                table = torch.tensor([[0, 0, O, I]], dtype=torch.int64, device=dev)
                _hip.call("snn_weight_frag_image_batched", src.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
                          9 * (I // 32) * (O // 32) * 128, flip, prec, st)
                return img
            img_f, img_b = image(w, Cout, Cin, 0, 4), image(wt, Cin, Cout, 1, 1)
            ops["fwd_halo"] = lambda: _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin, img_f.data_ptr(), y.data_ptr(), Cout, N,
                                                H, W, Cin, Cout, None, 0, None, 0, None, 0, None, 4, st)
            ops["dgrad_halo"] = lambda: _hip.call("snn_conv3x3_halo", dy.data_ptr(), Cout, img_b.data_ptr(), dx.data_ptr(), Cin,
                                                  N, H, W, Cout, Cin, None, 0, None, 0, None, 0, None, 1, st)
        for name, fn in ops.items():
            if args.only and not name.startswith(args.only):
                continue
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print(f"{label + f' {Cin}->{Cout} k{k}s{s}':28s} {name:6s} {us:9.1f} {flops / us / 1e6:8.1f}"
                  + (f"  splitk={splitk}" if name == "wgrad" else ""))


if __name__ == "__main__":
    main()
