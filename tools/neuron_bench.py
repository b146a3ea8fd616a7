#!/usr/bin/env python3
"""Micro-benchmark of the fused norm+neuron scans (forward / backward, with and without the BatchNorm sums), fp32 tensors
and the bf16-storage form (usage: neuron_bench.py [fp32|bf16|both])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import _hip  # noqa: E402
from snn_for_object_detection_amd.functional import neuron_params  # noqa: E402

_hip.load()
dev, st = torch.device("cuda"), torch.cuda.current_stream().cuda_stream
p = neuron_params()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for dt, flag, es in [d for d in ((torch.float32, 0, 4.0), (torch.bfloat16, _hip.SCAN_BF16_STORAGE, 2.0))
                     if which == "both" or (which == "bf16") == (d[1] != 0)]:
  for (T, B, H, W, C) in [(32, 5, 120, 152, 64), (32, 5, 120, 152, 32), (32, 5, 60, 76, 128), (32, 5, 60, 76, 64),
                          (32, 5, 30, 38, 256), (32, 5, 30, 38, 128), (32, 5, 15, 19, 128), (32, 5, 8, 10, 128)]:
      M = B * H * W
      y = torch.randn(T, M, C, device=dev).to(dt)
      go = torch.randn(T, M, C, device=dev).to(dt)
      vdec = (torch.randn(T, M, C, device=dev) + 0.5).to(dt)
      gx = torch.empty(T, M, C, device=dev, dtype=dt)
      out = torch.empty(T, M, C, device=dev, dtype=dt)
      alpha = torch.rand(T, C, device=dev) + 0.5
      beta = torch.randn(T, C, device=dev)
      nsum = _hip.query("snn_affine_neuron_bwd_sums_size", T, M, C)
      sums = torch.empty(nsum, device=dev, dtype=torch.float64)
      vT, iT = torch.empty(M, C, device=dev), torch.empty(M, C, device=dev)

      def fwd():
          _hip.call("snn_affine_neuron_fwd", 1, y.data_ptr(), C, alpha.data_ptr(), beta.data_ptr(), None, None,
                    out.data_ptr(), C, None, 0, vT.data_ptr(), iT.data_ptr(), vdec.data_ptr(), T, M, C, p, flag, st)

      def bwd(with_sums):
          _hip.call("snn_affine_neuron_bwd", 1, go.data_ptr(), C, vdec.data_ptr(), y.data_ptr(), C, None, None, None, None,
                    0, gx.data_ptr(), None, None, sums.data_ptr() if with_sums else None, T, M, C, p, flag, st)

      for name, fn, tensors in (("fwd", fwd, 3), ("bwd nosum", lambda: bwd(False), 3), ("bwd sums", lambda: bwd(True), 4)):
          for _ in range(2):
              fn()
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(10):
              fn()
          e1.record()
          torch.cuda.synchronize()
          us = 100.0 * e0.elapsed_time(e1)
          print(f"{'bf16' if flag else 'fp32'} T={T} M={M} C={C:4d} {name:10s} {us:8.1f} us  {tensors * es * T * M * C / us / 1e6:7.2f} TB/s")
