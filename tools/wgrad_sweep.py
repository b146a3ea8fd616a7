#!/usr/bin/env python3
"""Time snn_conv2d_wgrad for one shape under every tile variant / residency (tuning aid).
usage: wgrad_sweep.py Cin Cout k s H W [frames]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

Cin, Cout, k, s, H, W = map(int, sys.argv[1:7])
N = int(sys.argv[7]) if len(sys.argv) > 7 else 160
_hip.load()
dev, st, pad = torch.device("cuda"), torch.cuda.current_stream().cuda_stream, k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = (torch.rand(N, H, W, Cin, device=dev) < 0.1).float()
dy = torch.randn(N, Ho, Wo, Cout, device=dev)
dw = torch.empty(Cout, k, k, Cin, device=dev)
flops = 2.0 * N * Ho * Wo * Cout * k * k * Cin
BEST = os.environ.get("SWEEP_BEST") is not None   # print only the default and the three fastest variants
RESULTS = []


def run(label):
    splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, 1)
    ws = torch.empty(splitk * dw.numel(), device=dev)
    def call():
        _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Ho, Wo, Cout,
                  k, k, s, pad, 0, ws.data_ptr(), splitk, 1, st)
    for _ in range(2):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    RESULTS.append((us, label, splitk))
    if not BEST:
        print(f"{label:28s} splitk {splitk:5d}  {us:8.1f} us  {flops / us / 1e6:6.1f} TF", flush=True)
    return dw.clone()


run("warm-up")
RESULTS.clear()
ref = run("default")
for tile in range(6):
    for res in (2, 3, 4):
        os.environ["SNN_WGRAD_TILE"], os.environ["SNN_WGRAD_RESIDENT"] = str(tile), str(res)
        try:
            out = run(f"tile {tile} resident {res}")
            err = float((out - ref).abs().max() / ref.abs().max())
            if err > 1e-4:
                print("   MISMATCH", err)
        except RuntimeError as e:
            print(f"tile {tile} resident {res}: {e}")
if BEST:
    d = RESULTS[0]
    line = f"{Cin}->{Cout} k{k} s{s} {H}x{W}: default {d[0]:.1f} us (splitk {d[2]})"
    for us, label, sk in sorted(RESULTS[1:])[:3]:
        line += f" | {label} {us:.1f}"
    print(line)
