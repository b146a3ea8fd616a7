#!/usr/bin/env python3
"""Where does the training step synchronise the host with the GPU?  One TinyYolo GEN1 step under
torch.cuda.set_sync_debug_mode("warn") (every synchronising call warns with its Python stack), and the host time of the
step's phases (forward / loss / backward / step enqueue) against the GPU time of the step.

    python tools/sync_probe.py
"""
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snn_for_object_detection_amd as S  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    T, B, H, W = 32, 5, 240, 304
    g = torch.Generator().manual_seed(0)
    X = (torch.rand(T, B, 2, H, W, generator=g) < 0.05).float().to(dev)
    labels = torch.tensor([[[0, 0.2, 0.2, 0.5, 0.6], [1, 0.5, 0.4, 0.9, 0.8]]] * B, device=dev)
    torch.manual_seed(2)
    model = S.TinyYolo(num_classes=2, time_window=0).to(dev).train()
    tr = FlatTrainer(model, lr=1e-3)

    def step(times=None):
        t0 = time.perf_counter()
        tr.zero_grad()
        t1 = time.perf_counter()
        loss = model.training_step((X, labels))
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        tr.step()
        t4 = time.perf_counter()
        if times is not None:
            times.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # 1. synchronising calls
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        step()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print(f"synchronising calls in one step: {len(caught)}")
    for w in caught[:12]:
        print("  ", str(w.message)[:160], "@", w.filename.split("/")[-1], w.lineno)
    # 2. host time of the phases with the GPU idle at the start of each step (nothing to wait for but real syncs)
    times = []
    for _ in range(10):
        torch.cuda.synchronize()
        step(times)
    torch.cuda.synchronize()
    names = ("zero_grad", "forward+loss", "backward", "trainer.step")
    for k, n in enumerate(names):
        v = sorted(t[k] for t in times)
        print(f"host {n:14s} median {1e3 * v[len(v) // 2]:7.2f} ms")
    print(f"host total median {1e3 * sorted(sum(t) for t in times)[len(times) // 2]:.2f} ms")
    # 3. the same free-running (the host may run ahead)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"free-running: host enqueue {1e3 * t_enq / 20:.2f} ms/step, wall {1e3 * t_all / 20:.2f} ms/step")


if __name__ == "__main__":
    main()
