#!/usr/bin/env python3
"""Sustained, interleaved A/B of the halo-resident 3x3 kernel against the implicit GEMM (same process, same box):

    python tools/halo_ab.py [--rounds 12] [--iters 20]

Cold launches on this chip run 15-20 % slower than launches inside a busy stream (profiles/r02_clock_under_load.txt), and
boxes differ by more than that, so both kernels are timed alternately, `iters` launches at a time, after a 1.5 s warm-up
of back-to-back launches; min and median per kernel and direction."""
import argparse
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

SHAPES = [(160, 30, 38, 128, 128), (160, 60, 76, 64, 64), (160, 15, 19, 128, 128), (160, 8, 10, 128, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--co32", action="store_true", help="the 32-channel full-resolution layer only (A = direct 3x3 kernel)")
    args = ap.parse_args()
    if args.co32:
        SHAPES[:] = [(160, 120, 152, 32, 32), (160, 60, 76, 32, 32)]
    _hip.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    for N, H, W, Cin, Cout in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        wt = torch.empty(Cin, 3, 3, Cout, device=dev)
        y, dx = torch.empty(N, H, W, Cout, device=dev), torch.empty(N, H, W, Cin, device=dev)
        _hip.call("snn_weight_transpose", w.data_ptr(), wt.data_ptr(), Cout, 3, 3, Cin, st)

        def image(src, O, I, flip, prec):
            img = torch.empty(9 * O * I, device=dev)
            table = torch.tensor([[0, 0, O, I]], dtype=torch.int64, device=dev)
            _hip.call("snn_weight_frag_image_batched", src.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
                      9 * (I // 32) * (O // 32) * 128, flip, prec, st)
            return img
        img_f, img_b = image(w, Cout, Cin, 0, 4), image(wt, Cin, Cout, 1, 1)
        ops = {
            "fwd gather": lambda: _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W,
                                            Cin, H, W, Cout, 3, 3, 1, 1, None, 0, None, 0, None, 4, st),
            "fwd halo": lambda: _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin, img_f.data_ptr(), y.data_ptr(), Cout, N, H, W,
                                          Cin, Cout, None, 0, None, 0, None, 0, None, 4, st),
            "dgrad gather": lambda: _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N,
                                              H, W, Cin, H, W, Cout, 3, 3, 1, 1, None, 0, None, 0, 1, st),
            "dgrad halo": lambda: _hip.call("snn_conv3x3_halo", dy.data_ptr(), Cout, img_b.data_ptr(), dx.data_ptr(), Cin, N, H,
                                            W, Cout, Cin, None, 0, None, 0, None, 0, None, 1, st),
        }
        t0 = time.time()
        while time.time() - t0 < 1.5:      # warm-up: the clock the chip holds under sustained load
            for fn in ops.values():
                fn()
            torch.cuda.synchronize()
        times = {k: [] for k in ops}
        for _ in range(args.rounds):
            for name, fn in ops.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[name].append(1e3 * e0.elapsed_time(e1) / args.iters)
        flops = 2.0 * N * H * W * Cout * 9 * Cin
        print(f"--- N{N} {H}x{W} {Cin}->{Cout}")
        for name, t in times.items():
            print(f"{name:13s} min {min(t):7.1f} us  median {statistics.median(t):7.1f} us  {flops / statistics.median(t) / 1e6:6.1f} TF")


def stride2():
    """The one-pass stride-2 data gradient against the four-launch implicit GEMM (and the forward of the same layer)."""
    _hip.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    for N, H, W, Cin, Cout in [(160, 120, 152, 64, 128), (160, 60, 76, 128, 256), (160, 30, 38, 256, 256), (160, 15, 19, 256, 256)]:
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, Ho, Wo, Cout, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        wt = torch.empty(Cin, 3, 3, Cout, device=dev)
        y, dx = torch.empty(N, Ho, Wo, Cout, device=dev), torch.empty(N, H, W, Cin, device=dev)
        _hip.call("snn_weight_transpose", w.data_ptr(), wt.data_ptr(), Cout, 3, 3, Cin, st)
        img = torch.empty(9 * Cout * Cin, device=dev)
        table = torch.tensor([[0, 0, Cin, Cout]], dtype=torch.int64, device=dev)
        _hip.call("snn_weight_frag_image_batched", wt.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
                  9 * (Cout // 32) * (Cin // 32) * 128, 1, 1, st)
        ops = {
            "fwd gather": lambda: _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W,
                                            Cin, Ho, Wo, Cout, 3, 3, 2, 1, None, 0, None, 0, None, 4, st),
            "dgrad 4-pass": lambda: _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N,
                                              H, W, Cin, Ho, Wo, Cout, 3, 3, 2, 1, None, 0, None, 0, 1, st),
            "dgrad 1-pass": lambda: _hip.call("snn_conv3x3_s2_dgrad", dy.data_ptr(), Cout, img.data_ptr(), dx.data_ptr(), Cin, N,
                                              H, W, Cin, Ho, Wo, Cout, None, 0, None, 0, 1, st),
        }
        t0 = time.time()
        while time.time() - t0 < 1.0:
            for fn in ops.values():
                fn()
            torch.cuda.synchronize()
        times = {k: [] for k in ops}
        for _ in range(10):
            for name, fn in ops.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[name].append(1e3 * e0.elapsed_time(e1) / 10)
        flops = 2.0 * N * Ho * Wo * Cout * 9 * Cin
        print(f"--- stride 2: N{N} {H}x{W} {Cin}->{Cout}")
        for name, t in times.items():
            print(f"{name:13s} min {min(t):7.1f} us  median {statistics.median(t):7.1f} us  {flops / statistics.median(t) / 1e6:6.1f} TF")


if __name__ == "__main__":
    if "--stride2" in sys.argv:
        sys.argv.remove("--stride2")
        stride2()
    else:
        main()
