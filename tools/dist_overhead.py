#!/usr/bin/env python3
"""What the N > 1 collectives cost inside the step, measured with ONE rank on RCCL (no data moves between GPUs, so
whatever shows is plumbing: stream waits, RCCL kernel launches, work handles).

    python tools/dist_overhead.py [--steps 40]

Variants, interleaved in one process on one process group: no exchange (the N = 1 step); the full exchange (early all-reduce
from the backward hook + flag exchange + head all-reduce); the same without the overlap (one all-reduce in step()); the
same without the flag exchange.
"""
import argparse
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import snn_for_object_detection_amd as S  # noqa: E402
from snn_for_object_detection_amd.trainer import FlatTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if os.environ.get("SNN_WARM_STREAM") == "1":  # the weight-gradient side stream takes its hardware queue first
        from snn_for_object_detection_amd import functional as HF
        with torch.cuda.stream(HF._side_stream(dev)):
            torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
    if os.environ.get("SNN_INIT_FIRST") == "1":   # as bench.py does: the process group before anything else uses the GPU
        dist.init_process_group(os.environ.get("SNN_DIST_BACKEND", "nccl"), rank=0, world_size=1, device_id=dev)
    T, B, H, W = 32, 5, 240, 304
    g = torch.Generator().manual_seed(0)
    X = (torch.rand(T, B, 2, H, W, generator=g) < 0.05).float().to(dev)
    labels = torch.tensor([[[0, 0.2, 0.2, 0.5, 0.6], [1, 0.5, 0.4, 0.9, 0.8]]] * B, device=dev)
    torch.manual_seed(2)
    model = S.TinyYolo(num_classes=2, time_window=0).to(dev).train()

    variants = {
        "before init_process_group": dict(exchange_single_rank=False, _pre=True),
        "no exchange": dict(exchange_single_rank=False),
        "full exchange (overlapped)": dict(exchange_single_rank=True),
        "one all-reduce in step()": dict(exchange_single_rank=True, overlap_grad_exchange=False),
        "overlapped, no flag exchange": dict(exchange_single_rank=True, _no_flags=True),
        "one all-reduce, no flag exchange": dict(exchange_single_rank=True, overlap_grad_exchange=False, _no_flags=True),
    }
    if os.environ.get("SNN_QUICK") == "1":
        variants = {k: v for k, v in variants.items() if k in ("no exchange", "full exchange (overlapped)")}
    results = {k: [] for k in variants}
    for rnd in range(args.rounds):
        for name, kw in variants.items():
            if rnd > 0 and kw.get("_pre"):
                continue
            kw = dict(kw)
            no_flags = kw.pop("_no_flags", False)
            if not kw.pop("_pre", False) and not dist.is_initialized():
                dist.init_process_group(os.environ.get("SNN_DIST_BACKEND", "nccl"), rank=0, world_size=1, device_id=dev)
                if os.environ.get("SNN_TOUCH_COMM", "1") == "1":
                    dist.all_reduce(torch.ones(4, device=dev))
            model._snn_neck_grads_ready = None
            tr = FlatTrainer(model, lr=1e-3, **kw)
            if no_flags:
                tr._written_flags = lambda tr=tr: [s.written for s in tr.slots]

            def step():
                tr.zero_grad()
                loss = model.training_step((X, labels))
                loss.backward()
                tr.step()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            results[name].append(1e3 * (time.perf_counter() - t0) / args.steps)
            del tr
    for name, v in results.items():
        print(f"{name:36s} " + "  ".join(f"{x:7.3f}" for x in v) + " ms/step", file=sys.stderr)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
