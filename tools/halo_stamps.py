#!/usr/bin/env python3
"""Phase breakdown of k_conv_wgrad_halo from in-kernel cycle stamps (needs the -DSNN_TUNING library:
SNN_HIP_LIB=build/libsnn_hip_tuning.so).  usage: halo_stamps.py N H W Cin Cout stride"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402

N, H, W, Cin, Cout, s = (int(a) for a in sys.argv[1:7])
Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(N, H, W, Cin, device=dev)
dy = torch.randn(N, Ho, Wo, Cout, device=dev) * 1e-3
dw = torch.empty(Cout, 3, 3, Cin, device=dev)
splitk = _hip.query("snn_conv2d_wgrad_splitk", N, H, W, Cin, Ho, Wo, Cout, 3, 3, s, 1, 1)
ws = torch.empty(splitk, dw.numel(), device=dev)
for _ in range(3):
    _hip.call("snn_conv2d_wgrad", x.data_ptr(), Cin, dy.data_ptr(), Cout, dw.data_ptr(), N, H, W, Cin, Ho, Wo, Cout, 3, 3,
              s, 1, 0, ws.data_ptr(), splitk, 1, st)
torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * (1024 * 8))()
lib.snn_debug_halo_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.snn_debug_halo_stamps(buf, 1024 * 8)
a = np.array(buf[:], dtype=np.float64).reshape(1024, 8)
a = a[a[:, 5] > 0]
names = ["prologue", "K loops", "patch setup + loads issue + 2nd barrier", "1st barrier wait", "epilogue stores", "total"]
print(f"{Cin}->{Cout} s{s} {H}x{W} N={N}: blocks sampled {len(a)}, slabs {splitk}")
for i, n in enumerate(names):
    print(f"  {n:45s} median {np.median(a[:, i]):10.0f} cycles  ({100 * np.median(a[:, i]) / np.median(a[:, 5]):5.1f} %)")
