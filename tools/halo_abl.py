#!/usr/bin/env python3
"""Timing ablations of the halo-resident 3x3 kernel (csrc/conv_halo.hip) - which part of a block's life costs what.

    python -m snn_for_object_detection_amd._build --tuning
    SNN_HIP_LIB=build/libsnn_hip_tuning.so python tools/halo_abl.py

SNN_HALO_ABL selects a variant compiled into -DSNN_TUNING builds only (results are WRONG by construction):
1 no weight DMA, 2 no per-k-step wait + barrier, 4 no halo prefetch, 8 no output stores, 16 no halo fragment reads;
sums combine.  Variants are timed interleaved, several rounds, in one process (min and median)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

SHAPES = [(160, 30, 38, 128, 128), (160, 60, 76, 64, 64), (160, 15, 19, 128, 128)]
VARIANTS = [0, 1, 2, 4, 8, 16, 3, 7, 15, 31]


def main():
    _hip.load()
    dev = torch.device("cuda")
    st = torch.cuda.current_stream().cuda_stream
    for N, H, W, Cin, Cout in SHAPES:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        y = torch.empty(N, H, W, Cout, device=dev)
        img = torch.empty(9 * Cout * Cin, device=dev)
        table = torch.tensor([[0, 0, Cout, Cin]], dtype=torch.int64, device=dev)
        _hip.call("snn_weight_frag_image_batched", w.data_ptr(), img.data_ptr(), table.data_ptr(), 1,
                  9 * (Cin // 32) * (Cout // 32) * 128, 0, 4, st)
        flops = 2.0 * N * H * W * Cout * 9 * Cin

        def run():
            _hip.call("snn_conv3x3_halo", x.data_ptr(), Cin, img.data_ptr(), y.data_ptr(), Cout, N, H, W, Cin, Cout, None,
                      0, None, 0, None, 0, None, 4, st)

        times = {v: [] for v in VARIANTS}
        for rnd in range(6):
            for v in VARIANTS:
                os.environ["SNN_HALO_ABL"] = str(v)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(1e3 * e0.elapsed_time(e1) / 5)
        print(f"--- N{N} {H}x{W} {Cin}->{Cout}")
        for v in VARIANTS:
            t = times[v][1:]
            print(f"abl {v:2d}: min {min(t):7.1f} us  median {statistics.median(t):7.1f} us  "
                  f"{flops / min(t) / 1e6:6.1f} TF (algorithmic, as if complete)")
    os.environ.pop("SNN_HALO_ABL", None)


if __name__ == "__main__":
    main()
