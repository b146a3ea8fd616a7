#!/usr/bin/env python3
"""Phase timing of the pipelined conv loop (needs a -DSNN_STAMP scratch build; SNN_HIP_LIB points at it).
usage: stamp_conv.py fwd|dgrad Cin Cout k s H W [frames]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

op, Cin, Cout, k, s, H, W = sys.argv[1], *map(int, sys.argv[2:8])
N = int(sys.argv[8]) if len(sys.argv) > 8 else 160
lib = _hip.load()
dev, st, pad = torch.device("cuda"), torch.cuda.current_stream().cuda_stream, k // 2
Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) * 0.05
wt = torch.randn(Cin, k, k, Cout, device=dev) * 0.05
y = torch.empty(N, Ho, Wo, Cout, device=dev)
dy = torch.randn(N, Ho, Wo, Cout, device=dev)
dx = torch.empty_like(x)
for _ in range(3):
    if op == "fwd":
        _hip.call("snn_conv2d_fwd", x.data_ptr(), Cin, w.data_ptr(), None, y.data_ptr(), Cout, N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, None, 4, st)
    else:
        _hip.call("snn_conv2d_dgrad", dy.data_ptr(), Cout, wt.data_ptr(), None, dx.data_ptr(), Cin, N, H, W, Cin, Ho, Wo, Cout, k, k, s, pad, None, 0, None, 0, 1, st)
torch.cuda.synchronize()
nblk = min(2048, (N * (H if op == "dgrad" else Ho) * (W if op == "dgrad" else Wo) + 127) // 128)
nblk = nblk // 8 * 8  # XCD-aware order: the last ids may be padding blocks
buf = np.zeros(2048 * 8, dtype=np.uint64)
lib.snn_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.snn_debug_stamps(buf.ctypes.data, 2048 * 8)
assert rc == 0, rc
a = buf.reshape(2048, 8)[:nblk].astype(np.float64)
stages = (k * k * (Cout if op == "dgrad" else Cin)) // 32
names = ["mfma0+convA", "mfma1+convB", "issue loads", "barrier 1", "lds writes", "barrier 2", "loop total"]
buf2 = np.zeros(2048 * 4, dtype=np.uint64)
lib.snn_debug_stamps2.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.snn_debug_stamps2(buf2.ctypes.data, 2048 * 4) == 0
b = buf2.reshape(2048, 4)[:nblk].astype(np.float64)
t0 = b[:, 0].min()
print(f"kernel span {b[:, 1].max() - t0:.0f} cycles; block lifetime mean {np.mean(b[:, 1] - b[:, 0]):.0f}; prologue mean "
      f"{np.mean(a[:, 7] - b[:, 0]):.0f}; epilogue mean {np.mean(b[:, 1] - a[:, 7] - a[:, 6]):.0f}")
order = np.argsort(b[:, 0])
print("block start times (sorted, every 100th):", (b[order, 0] - t0)[::100].astype(int).tolist())
start = a[:, 7] - a[:, 7].min()
first = start < 2000  # blocks of the first resident round
print(f"{nblk} blocks, {stages} k-steps; first-round blocks {first.sum()}")
for sel, label in ((first, "first round"), (~first, "later rounds")):
    if sel.sum() == 0:
        continue
    print(f" {label}: per k-step cycles (mean over blocks)")
    for i, n in enumerate(names):
        print(f"   {n:14s} {a[sel, i].mean() / stages:9.1f}")
