#!/usr/bin/env python3
"""Headline benchmark: event-frames/s (B x T) of one SODa / TinyYolo training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the whole hot path on one batch of synthetic GEN1-shaped events (BASELINE.json configs[1]:
SODa 3M / TinyYolo, 304x240, B=5, T=32 per GPU): layer-major forward over all T, detection loss on
the last timestep, BPTT backward producing every parameter gradient, (N>1: one RCCL all-reduce of the
flat gradient), fused Adamax update.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  Beside the contract fields it carries
  "roofline":     the dominant kernel's achieved rate (HIP events on the launch stream, algorithmic
                  FLOPs of SURVEY 8(d)) against the MI355X peak,
  "cpu_baseline": the CPU oracle (pure-PyTorch restatement of the reference, time-outer loop) timed
                  on this host on a bounded sample of the same workload (rank 0, N=1 only).
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GEN1_H, GEN1_W = 240, 304
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def mfma_peak_for(kernel: str, fwd_prec: str, bwd_prec: str):
    """Peak ALGORITHMIC TFLOP/s of a conv kernel: the split-precision kernels spend 3 (bf16x3) or 6 (bf16x6) dense
    bf16 MFMA products per algorithmic multiply-add, the exact kernels one fp32 MFMA product."""
    backward = (kernel.startswith("k_conv_wgrad") or ", true, " in kernel  # k_conv_gather<BN, WM, WN, DGRAD, VEC>
                or kernel.endswith(", true>"))  # k_conv_direct3<BN, WM, WN, DGRAD>
    prec = bwd_prec if backward else fwd_prec
    if prec == "bf16x3":
        return PEAK_BF16_MATRIX_TFLOPS / 3.0, "bf16 dense MFMA / 3 products"
    if prec == "bf16x6":
        return PEAK_BF16_MATRIX_TFLOPS / 6.0, "bf16 dense MFMA / 6 products"
    if prec == "fp16x3":
        return PEAK_BF16_MATRIX_TFLOPS / 3.0, "fp16 dense MFMA / 3 products"
    return PEAK_F32_MATRIX_TFLOPS, "fp32 MFMA"


def synthetic_batch(T, B, H, W, num_classes, device, seed):
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(T, B, 2, H, W, generator=g) < 0.05).float()
    labels = torch.full((B, 2, 5), -1.0)
    for b in range(B):
        for k in range(2):
            while True:
                xy = torch.rand(2, 2, generator=g)
                lo, hi = xy.min(0).values, xy.max(0).values
                if (hi - lo).prod() > 0.01:
                    break
            labels[b, k, 0] = float(torch.randint(0, num_classes, (1,), generator=g))
            labels[b, k, 1:3], labels[b, k, 3:5] = lo, hi
    return X.to(device), labels.to(device)


def cpu_baseline(sample_T, sample_B, H, W, num_classes, threads):
    """Time the oracle (port of the reference path) on the host cores: 1 warm-up + 2 timed steps."""
    from oracle.net import SODaRef
    import snn_for_object_detection_amd as S
    torch.set_num_threads(threads)
    torch.manual_seed(2)
    desc = S.TinyYolo(num_classes=num_classes, time_window=0)
    model = SODaRef(desc, num_classes, time_window=0)
    model.load_state_dict(desc.state_dict())
    model.train()
    X, labels = synthetic_batch(sample_T, sample_B, H, W, num_classes, "cpu", seed=0)
    times = []
    for it in range(3):
        model.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss = model.training_step((X, labels))
        loss.backward()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {
        "value": sample_T * sample_B / best,
        "unit": "event-frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle (pure-PyTorch fp32 restatement, time-outer loop; norse not installed) fwd+bwd on "
                  f"TinyYolo GEN1 {W}x{H} B={sample_B} T={sample_T}, best of 2 after 1 warm-up, {best:.2f} s/step",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=5, help="samples per GPU")
    ap.add_argument("--timesteps", type=int, default=32)
    ap.add_argument("--height", type=int, default=GEN1_H)
    ap.add_argument("--width", type=int, default=GEN1_W)
    ap.add_argument("--classes", type=int, default=2)
    ap.add_argument("--sync-bn", action="store_true",
                    help="SyncBatchNorm as in the reference's config.yaml:76 (off by default, N>1 only)")
    ap.add_argument("--forward-precision", choices=("fp16x3", "bf16x6", "fp32"), default="fp16x3",
                    help="forward conv arithmetic: fp16x3 = 2 fp16 pieces, 3 products (fp32-grade for |x| < 4094); "
                         "bf16x6 = 3 bf16 pieces, 6 products (fp32-grade, any range); fp32 = exact fp32 MFMA")
    ap.add_argument("--backward-precision", choices=("bf16x3", "fp32"), default="bf16x3",
                    help="arithmetic of the backward convolutions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table to stderr")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") over xGMI is the production transport; SNN_DIST_BACKEND=gloo exists only to rehearse the
        # multi-process control flow on a box with fewer GPUs than ranks
        backend = os.environ.get("SNN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    import snn_for_object_detection_amd as S
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.profiler import KernelProfiler
    from snn_for_object_detection_amd.trainer import FlatTrainer, broadcast_parameters
    _hip.load()
    S.functional.set_forward_precision(args.forward_precision)
    S.functional.set_backward_precision(args.backward_precision)

    T, B, H, W = args.timesteps, args.batch, args.height, args.width
    torch.manual_seed(2)  # same reference init on every rank
    model = S.TinyYolo(num_classes=args.classes, time_window=0).to(device).train()
    trainer = FlatTrainer(model, lr=model.hparams.learning_rate)
    broadcast_parameters(trainer)
    if args.sync_bn and world > 1:
        from snn_for_object_detection_amd.trainer import convert_sync_batchnorm
        convert_sync_batchnorm(model)
    X, labels = synthetic_batch(T, B, H, W, args.classes, device, seed=rank)  # a different shard per rank

    def step():
        trainer.zero_grad()
        loss = model.training_step((X, labels))
        loss.backward()
        trainer.step()
        return loss

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    frames_per_s = world * B * T * args.steps / elapsed

    roofline = None
    if not args.no_roofline:
        # every rank runs the two extra steps (they contain the all-reduce); only rank 0 brackets its launches
        # kernels are timed one at a time: the weight-gradient side stream is switched off for these two steps
        from snn_for_object_detection_amd import functional as HF
        prof = KernelProfiler() if rank == 0 else None
        _hip.PROFILER = prof
        HF.USE_WGRAD_STREAM = False
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        HF.USE_WGRAD_STREAM = True
        _hip.PROFILER = None
    if rank == 0 and not args.no_roofline:
        table = prof.summary()
        total_ms = sum(r["ms"] for r in table.values())
        name, row = max(table.items(), key=lambda kv: kv[1]["ms"])
        # HBM bytes per launch of that kernel: rocprofv3 PMC passes of this same workload (FETCH_SIZE x2 per the
        # gfx950 correction, + WRITE_SIZE; tools/pmc_traffic.py), committed under profiles/ - bench.py cannot
        # run the profiler on itself, so the field is null when no such file is present
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath) and (T, B, H, W) == (32, 5, GEN1_H, GEN1_W):
            traffic = json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
        peak, peak_note = mfma_peak_for(name, args.forward_precision, args.backward_precision)
        roofline = {
            "bound": "mfma", "kernel": name, "achieved": row["tflops"], "peak": peak, "peak_basis": peak_note,
            "unit": "TFLOP/s", "frac": row["tflops"] / peak, "traffic": traffic,
            "achieved_hbm_gbs": row["gbs"], "frac_hbm": row["gbs"] / PEAK_HBM_GBS,
            "avg_launch_us": row["avg_us"], "launches_per_step": row["calls"] // 2,
            "share_of_kernel_time": row["ms"] / total_ms,
            "flops_per_launch": row["flops"] / row["calls"],
            "all_kernels_ms_per_step": total_ms / 2,
        }
        if args.kernel_table:
            for k, r in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"[kernel] {k:48s} calls/step {r['calls'] // 2:4d}  avg {r['avg_us']:9.1f} us  "
                      f"{100 * r['ms'] / total_ms:5.1f}%  {r['tflops']:7.2f} TFLOP/s  {r['gbs']:8.1f} GB/s",
                      file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(os.cpu_count() or 1, 16)  # the GPU box's CPU share for one GPU
        # bounded sample of the same workload: the full batch of 5, half the timesteps (about 10 s of CPU work, 26 GiB)
        cpu = cpu_baseline(sample_T=min(T, 16), sample_B=B, H=H, W=W, num_classes=args.classes, threads=threads)

    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "event-frames/sec (BxT) SODa-3M GEN1 304x240 fwd+bwd",
            "value": frames_per_s,
            "unit": "event-frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"SODa/TinyYolo (4.23M params) GEN1 {W}x{H}, B={B}/GPU T={T}, p(event)=0.05, 2 boxes/sample, "
                            "fwd + loss(last step) + BPTT bwd + flat-grad all-reduce (N>1) + fused Adamax",
                "global_batch": B * world, "timesteps": T, "parallelism": f"dp{world}",
                "sync_batchnorm": bool(args.sync_bn and world > 1),
                "arithmetic": "fp32 storage and accumulation; forward conv "
                              + {"fp16x3": "fp16x3 split products (two fp16 pieces per operand after exact 2^k "
                                           "pre-scaling, hh+hl+lh; fp32-grade: rel 5e-7 vs fp64, same as the fp32 MFMA)",
                                 "bf16x6": "bf16x6 split products (3-way bf16 split of both operands, fp32-grade: "
                                           "rel 5e-7 vs fp64, same as the fp32 MFMA)",
                                 "fp32": "exact fp32 MFMA"}[args.forward_precision]
                              + "; backward conv "
                              + ("bf16x3 split products (hi*hi+hi*lo+lo*hi, rel 1e-5)"
                                 if args.backward_precision == "bf16x3" else "exact fp32 MFMA"),
                "loss": float(loss.item()),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
