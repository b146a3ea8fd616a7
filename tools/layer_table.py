#!/usr/bin/env python3
"""Per-shape kernel table of one TinyYolo GEN1 training step (HIP events around every C-ABI call).

    python tools/layer_table.py [--batch 5] [--timesteps 32] [--top 60]

Groups the calls of one step by (entry point, shape) so the expensive LAYERS are visible, not just the
expensive kernel families (bench.py --kernel-table).  The weight-gradient side stream is switched off so
every call is timed in isolation.
"""
import argparse
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snn_for_object_detection_amd import TinyYolo, _hip, functional  # noqa: E402
from snn_for_object_detection_amd.profiler import KernelProfiler, work_of  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402


def shape_of(name, a):
    if name in ("snn_conv2d_fwd", "snn_conv2d_dgrad"):
        a = tuple(a[:3]) + tuple(a[4:])   # without the pre-split weight pointer: the positions snn_conv2d_wgrad has
    if name == "snn_conv2d_wgrad_bn":   # (x, gx, y, the four coefficient rows, ... first: geometry starts at position 10)
        return f"N{a[10]} {a[11]}x{a[12]} {a[13]}->{a[16]} k{a[17]} s{a[19]} +bn-apply"
    if name.startswith("snn_conv2d"):
        return f"N{a[5]} {a[6]}x{a[7]} {a[8]}->{a[11]} k{a[12]} s{a[14]}"
    if name == "snn_conv3x3_halo":
        return f"N{a[5]} {a[6]}x{a[7]} {a[8]}->{a[9]} {'fwd' if a[17] == 4 else 'dgrad'}{' +add' if a[10] is not None else ''}{' +add2' if a[12] is not None else ''}"
    if name == "snn_conv3x3_s2_dgrad":
        return f"N{a[5]} {a[6]}x{a[7]} {a[8]}->{a[11]}"
    if name == "snn_affine_neuron_fwd":
        return f"n{a[0]} T{a[14]} M{a[15]} C{a[16]}{' +shortcut' if a[9] is not None else ''}"
    if name == "snn_affine_neuron_bwd":
        return f"n{a[0]} T{a[15]} M{a[16]} C{a[17]}"
    if name == "snn_add":
        return f"M{a[6]} C{a[7]}"
    if name == "snn_bn_bwd_apply":
        return f"T{a[8]} M{a[9]} C{a[10]}"
    if name == "snn_bn_stats":
        return f"T{a[2]} M{a[3]} C{a[4]}"
    return ""


class ShapeProfiler(KernelProfiler):
    def before(self, name, args):
        tok = super().before(name, args)
        return (f"{name} {shape_of(name, args)}",) + tok[1:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=5)
    ap.add_argument("--timesteps", type=int, default=32)
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = TinyYolo(num_classes=2, time_window=0).to(dev).train()
    trainer = FlatTrainer(model)
    X = (torch.rand(args.timesteps, args.batch, 2, 240, 304, device=dev) < 0.05).float()
    labels = torch.tensor([[[0, 0.2, 0.2, 0.5, 0.6], [1, 0.5, 0.4, 0.9, 0.8]]] * args.batch, device=dev)
    functional.USE_WGRAD_STREAM = False

    def step():
        trainer.zero_grad()
        loss = model.training_step((X, labels), 0)
        loss.backward()
        trainer.step()

    for _ in range(2):
        step()
    prof = ShapeProfiler()
    _hip.PROFILER = prof
    step()
    _hip.PROFILER = None
    rows = prof.summary()
    total = sum(r["ms"] for r in rows.values())
    print(f"total {total:.2f} ms over {sum(r['calls'] for r in rows.values())} calls")
    for label, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"])[: args.top]:
        print(f"{label:58s} x{r['calls']:<3d} {r['ms']:7.3f} ms {100 * r['ms'] / total:5.1f}%  avg {r['avg_us']:8.1f} us"
              f" {r['tflops']:7.1f} TF {r['gbs']:7.0f} GB/s")


if __name__ == "__main__":
    main()
