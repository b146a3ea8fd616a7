#!/usr/bin/env python3
"""Headline benchmark: event-frames/s (B x T) of one SODa / TinyYolo training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config gen1|1mpx|deep12]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the whole hot path on one batch of synthetic events: layer-major forward over all T, loss on the
last timestep, BPTT backward producing every parameter gradient, (N>1: one RCCL all-reduce of the flat
gradient), fused Adamax update.  Inputs are resident in HBM before the timed region.

Workloads (``--config``; per-GPU batch, weak scaling):
  gen1   (default, the headline: BASELINE.json configs[1] / [2]) SODa 3M / TinyYolo, GEN1 304x240, B=5, T=32
  1mpx   BASELINE.json configs[3]: TinyYolo, 1 Mpx 1280x720, 7 classes, B=8, T=32 (about 250 GiB of HBM)
  deep12 BASELINE.json configs[4]: 12 x {Conv(64,3), Norm, LIF} on 304x240, T=128, B=2 per GPU (16 on 8 GPUs)

Prints ONE JSON line (rank 0).  Beside the contract fields it carries
  "roofline":     the dominant kernel's achieved rate (HIP events on the launch stream, algorithmic
                  FLOPs of SURVEY 8(d)) against the MI355X peak,
  "cpu_baseline": the CPU oracle (pure-PyTorch restatement of the reference, time-outer loop) timed
                  on this host on a bounded sample of the same workload (rank 0, N=1 only),
  "dist":         the torch.distributed backend that carried the gradient all-reduce and the ranks it saw.
"""

import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GEN1_H, GEN1_W = 240, 304
PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MATRIX_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA
PEAK_HBM_GBS = 8000.0

# name -> (H, W, classes, per-GPU batch, T, event density, CPU-baseline sample (B, T))
CONFIGS = {
    "gen1": dict(H=GEN1_H, W=GEN1_W, classes=2, batch=5, T=32, p=0.05, cpu_sample=(5, 32),
                 label="SODa/TinyYolo (4.23M params) GEN1 304x240"),
    "1mpx": dict(H=720, W=1280, classes=7, batch=8, T=32, p=0.05, cpu_sample=(1, 4),
                 label="SODa/TinyYolo (4.26M params) 1Mpx 1280x720, 7 classes"),
    "deep12": dict(H=GEN1_H, W=GEN1_W, classes=0, batch=2, T=128, p=0.3, cpu_sample=(1, 8),
                   label="deep backbone 12 x {Conv(64,3), Norm, LIF} on GEN1 304x240"),
}
# significant bits of one product of each convolution arithmetic (include/snn_hip.h, SNN_PREC_*)
PRODUCT_BITS = {"fp32": 24, "fp16x3": 22, "bf16x6": 24, "bf16x3": 16, "bf16": 8}


def mfma_peak_for(kernel: str, fwd_prec: str, bwd_prec: str):
    """Peak ALGORITHMIC TFLOP/s of a conv kernel: the split-precision kernels spend 3 (bf16x3 / fp16x3) or 6 (bf16x6)
    dense 16-bit MFMA products per algorithmic multiply-add, the exact kernels one fp32 MFMA product."""
    backward = (kernel.startswith("k_conv_wgrad") or "wgrad" in kernel or ", true, " in kernel  # k_conv_gather<BN, WM, WN, DGRAD, VEC>
                or kernel.endswith(", true>")     # k_conv_direct3<BN, WM, WN, DGRAD>
                or "dgrad" in kernel)             # k_conv_halo3<CO, dgrad>, k_conv_s2dgrad3
    prec = bwd_prec if backward else fwd_prec
    if "spikes" in kernel:   # operand thresholded from saved potentials: one exact piece, two products (default arithmetic only)
        return PEAK_BF16_MATRIX_TFLOPS / 2.0, "16-bit dense MFMA / 2 products (spike operand: one exact piece)"
    if "bf16s" in kernel:   # bf16-storage mode: stored bf16 activations x bf16-rounded weights, one product
        return PEAK_BF16_MATRIX_TFLOPS, "bf16 dense MFMA, one product (bf16 storage)"
    if prec == "bf16":
        return PEAK_BF16_MATRIX_TFLOPS, "bf16 dense MFMA, one product"
    if prec == "bf16x3":
        return PEAK_BF16_MATRIX_TFLOPS / 3.0, "bf16 dense MFMA / 3 products"
    if prec == "bf16x6":
        return PEAK_BF16_MATRIX_TFLOPS / 6.0, "bf16 dense MFMA / 6 products"
    if prec == "fp16x3":
        return PEAK_BF16_MATRIX_TFLOPS / 3.0, "fp16 dense MFMA / 3 products"
    return PEAK_F32_MATRIX_TFLOPS, "fp32 MFMA"


def roofline_head(name, row, fwd_prec, bwd_prec, sb):
    """Which roof binds a kernel family is decided by its ARITHMETIC INTENSITY (algorithmic FLOPs / algorithmic bytes of
    its launches) against the ridge of its own peak pair - matrix peak of its arithmetic / 8 TB/s - not by its
    name: a convolution family whose intensity lies below the ridge (the weight gradients of a step average 70 FLOP/B
    against a ridge of 104) is HBM-bound.  Both fractions stay in the line (``frac_mfma`` / ``frac_hbm``)."""
    hbm_basis = (f"HBM3E 8 TB/s spec (6.3 TB/s measured copy rate); algorithmic bytes = {2 if sb else 4} B per "
                 "tensor element the kernel must touch")
    if not name.startswith("k_conv") or row["bytes"] <= 0:
        # the fused norm + neuron scans and the other pointwise kernels move bytes: priced against HBM
        return {"bound": "hbm", "kernel": name, "achieved": row["gbs"], "peak": PEAK_HBM_GBS, "peak_basis": hbm_basis,
                "unit": "GB/s", "frac": row["gbs"] / PEAK_HBM_GBS}
    peak, peak_note = mfma_peak_for(name, fwd_prec, bwd_prec)
    intensity = row["flops"] / row["bytes"]
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)
    common = {"kernel": name, "intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
              "frac_mfma": row["tflops"] / peak, "mfma_peak": peak, "mfma_peak_basis": peak_note}
    if intensity < ridge:
        return {"bound": "hbm", **common, "achieved": row["gbs"], "peak": PEAK_HBM_GBS,
                "peak_basis": hbm_basis + f"; the family's intensity {intensity:.0f} FLOP/B is below the ridge "
                                          f"{ridge:.0f} of {peak_note}",
                "unit": "GB/s", "frac": row["gbs"] / PEAK_HBM_GBS}
    return {"bound": "mfma", **common, "achieved": row["tflops"], "peak": peak, "peak_basis": peak_note,
            "unit": "TFLOP/s", "frac": row["tflops"] / peak}


def chain_row(table, neuron_steps, profiled_steps, sb):
    """The Norm + neuron CHAIN as one row: forward scan + reverse scan + BatchNorm-backward apply time against SURVEY
    8(d)'s ideal-fusion bytes - 5 tensors per neuron-timestep (fwd: read y, write z; bwd: read dz, read y, write dy) =
    20 B with fp32 tensors, 10 B with bf16 - NOT the bytes each of today's kernels moves (the per-kernel rows)."""
    keys = [k for k in table if k.startswith(("k_affine_neuron_fwd", "k_affine_neuron_bwd", "k_bn_bwd_apply", "k_lif_bwd"))]
    ms = sum(table[k]["ms"] for k in keys)
    moved = sum(table[k]["bytes"] for k in keys)
    if ms <= 0 or neuron_steps <= 0:
        return None
    per = 10.0 if sb else 20.0
    ideal = per * neuron_steps
    gbs = ideal / (ms * 1e-3) / 1e9
    return {"kernel": "chain: Norm + neuron (fwd scan + bwd scan + BN-backward apply)", "members": sorted(keys),
            "ms_per_step": ms / profiled_steps, "neuron_timesteps_per_step": neuron_steps / profiled_steps,
            "ideal_bytes_per_neuron_timestep": per, "moved_bytes_per_neuron_timestep": moved / neuron_steps,
            "traffic_ratio": moved / ideal, "gbs": gbs, "frac_hbm": gbs / PEAK_HBM_GBS,
            "basis": "SURVEY 8(d): fused Norm+LIF kernel alone = 5 x s bytes per neuron-timestep"}


def wgrad_family_row(table, total_ms, profiled_steps):
    """All weight-gradient kernels as ONE row (what rounds 1-3 called the k_conv_wgrad family): the implicit GEMM of the 1x1
    layers (HBM-bound), the halo-resident 3x3 kernel (matrix-bound) and the event-frame row kernel are priced separately
    above; the sum keeps the round-over-round comparison."""
    keys = [k for k in table if k.split(",")[0] in ("k_conv_wgrad_pipe", "k_conv_wgrad_halo", "k_conv_first<wgrad>")]
    ms = sum(table[k]["ms"] for k in keys)
    if ms <= 0:
        return None
    flops, byts, calls = (sum(table[k][f] for k in keys) for f in ("flops", "bytes", "calls"))
    sec = ms * 1e-3
    return {"kernel": "k_conv_wgrad (all weight-gradient kernels + their ordered reduce)", "members": sorted(keys),
            "ms_per_step": ms / profiled_steps, "share": ms / total_ms, "launches_per_step": calls // profiled_steps,
            "tflops": flops / sec / 1e12, "gbs": byts / sec / 1e9, "frac_hbm": byts / sec / 1e9 / PEAK_HBM_GBS,
            "frac_mfma": flops / sec / 1e12 / (PEAK_BF16_MATRIX_TFLOPS / 3.0)}


def synthetic_batch(T, B, H, W, num_classes, device, seed, p=0.05):
    g = torch.Generator().manual_seed(seed)
    X = (torch.rand(T, B, 2, H, W, generator=g) < p).float()
    labels = torch.full((B, 2, 5), -1.0)
    for b in range(B):
        for k in range(2):
            while True:
                xy = torch.rand(2, 2, generator=g)
                lo, hi = xy.min(0).values, xy.max(0).values
                if (hi - lo).prod() > 0.01:
                    break
            labels[b, k, 0] = float(torch.randint(0, max(num_classes, 1), (1,), generator=g))
            labels[b, k, 1:3], labels[b, k, 3:5] = lo, hi
    return X.to(device), labels.to(device)


def deep12_cfg():
    from snn_for_object_detection_amd import Conv, LIF, Norm
    cfg = []
    for _ in range(12):
        cfg += [Conv(64, 3), Norm(), LIF()]
    return cfg


def deep12_probe(B, H, W, device):
    """Read-out of the backbone at the last timestep: a fixed random projection (the gradient then flows back through
    all T steps of all 12 layers)."""
    return torch.randn(B, 64, H, W, generator=torch.Generator().manual_seed(3)).to(device)


def csrc_fingerprint() -> str:
    """sha256 over the kernel sources: ties a PMC traffic file to the kernels it was measured on (the digest the build
    stamps ``libsnn_hip.so`` with, ``_build.source_fingerprint``)."""
    from snn_for_object_detection_amd._build import source_fingerprint
    return source_fingerprint()


def pmc_traffic(config: str, kernel: str):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes of this workload
    (tools/pmc_traffic.py; FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE).  bench.py cannot run the profiler
    on itself, so the number comes from profiles/ - and only while that file was measured on the CURRENT kernel
    sources; otherwise the field is null and the provenance says why."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{config}.json")))   # newest round last
    if not cands:
        return None, {"file": None, "note": "no PMC traffic file for this workload under profiles/"}
    path = cands[-1]
    name = os.path.basename(path)
    data = json.load(open(path))
    meta = data.get("_meta", {})
    now = csrc_fingerprint()
    src = {"file": f"profiles/{name}", "file_sha256_16": hashlib.sha256(open(path, "rb").read()).hexdigest()[:16],
           "measured_on_csrc": str(meta.get("csrc_sha256", "?"))[:16], "current_csrc": now[:16]}
    if meta.get("csrc_sha256") != now:
        src["note"] = "stale: kernel sources changed since the PMC passes; traffic nulled"
        return None, src
    # (the counter passes file a spike-operand instance under its plain kernel family: same program text, one product less)
    return data.get(kernel.replace(", spikes", ""), {}).get("hbm_bytes_per_launch"), src


def dominant_template(prof):
    """The single kernel TEMPLATE (entry point + shape class) with the most time inside the profiled steps, beside the
    family the roofline is quoted for: {"kernel", "ms_per_step", "tflops", "gbs", "launches_per_step"}."""
    agg = {}
    for label, flops, byts, start, end in prof.records:
        row = agg.setdefault(label, [0.0, 0.0, 0.0, 0])
        row[0] += start.elapsed_time(end)
        row[1] += flops
        row[2] += byts
        row[3] += 1
    label, (ms, flops, byts, calls) = max(agg.items(), key=lambda kv: kv[1][0])
    sec = max(ms, 1e-9) * 1e-3
    return {"kernel": label, "ms_per_step": ms / 2, "tflops": flops / sec / 1e12, "gbs": byts / sec / 1e9,
            "launches_per_step": calls // 2}


def host_cpu():
    """CPU model and core counts of this host (printed beside the baseline)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count()
    return {"cpu_model": model, "host_logical_cores": os.cpu_count(), "cores_available_to_process": usable}


def cpu_baseline(config, H, W, num_classes, threads, p):
    """Time the oracle (port of the reference path) on the host cores: 1 warm-up + 2 timed steps."""
    from oracle.net import BlockRef, SODaRef
    import snn_for_object_detection_amd as S
    sample_B, sample_T = CONFIGS[config]["cpu_sample"]
    torch.set_num_threads(threads)
    torch.manual_seed(2)
    X, labels = synthetic_batch(sample_T, sample_B, H, W, num_classes, "cpu", seed=0, p=p)
    if config == "deep12":
        model = BlockRef(2, deep12_cfg()).train()
        probe = deep12_probe(sample_B, H, W, "cpu")

        def run():
            state, out = None, None
            for t in range(sample_T):
                out, state = model(X[t], state)
            return (out * probe).mean()
    else:
        desc = S.TinyYolo(num_classes=num_classes, time_window=0)
        model = SODaRef(desc, num_classes, time_window=0)
        model.load_state_dict(desc.state_dict())
        model.train()

        def run():
            return model.training_step((X, labels))
    times = []
    for it in range(3):
        model.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        loss = run()
        loss.backward()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {
        "value": sample_T * sample_B / best,
        "unit": "event-frames/s",
        "cores": threads,
        **host_cpu(),
        "kind": "port",
        "sample": f"oracle (pure-PyTorch fp32 restatement, time-outer loop; norse not installed) fwd+bwd on "
                  f"{CONFIGS[config]['label']} B={sample_B} T={sample_T}, best of 2 after 1 warm-up, {best:.2f} s/step",
    }


def spawn_ranks(n: int) -> int:
    """Run this script as ``n`` ranks of one node under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1) and return the launcher's exit code; rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd)


_JSON_FD = None


def reserve_stdout() -> None:
    """Keep file descriptor 1 for the ONE JSON line: everything else any library writes to stdout during the run goes to
    stderr instead (RCCL prints a five-line version banner on stdout when its communicator comes up - rank 0's stdout
    then no longer parses as a JSON line)."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj) -> None:
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    if _JSON_FD is None:
        os.write(1, line)
    else:
        os.write(_JSON_FD, line)


def build_model(args, cfg, classes, device):
    """-> (model, lr): the workload's network with the reference initialisation (same seed on every rank)."""
    import snn_for_object_detection_amd as S
    torch.manual_seed(2)
    if args.config == "deep12":
        model = S.BlockGen(2, deep12_cfg())
        for m in model.modules():
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        return model.to(device).train(), 1e-3
    model = S.TinyYolo(num_classes=classes, time_window=0).to(device).train()
    return model, model.hparams.learning_rate


def replicas_equal(trainer, dist, what="flat_param") -> bool:
    """Every rank holds the same buffer (ranks see different shards: a missing or partial exchange shows here)."""
    chk = getattr(trainer, what).double().abs().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool((lo == hi).item())


def dry_run(args, world: int, rank: int) -> int:
    """The multi-process control flow of the N > 1 step without a GPU: rendezvous, the REAL model and ``FlatTrainer`` of the
    workload on the CPU, and per step every collective the real step issues, in the order it issues them - weight
    broadcast, (``--sync-bn``) one ``[T, C, 2]`` fp64 all-reduce per BatchNorm layer forward and one backward, the early
    neck + head gradient all-reduce from the backward hook, the backbone part + join in ``all_reduce()`` - over fake
    rank-specific gradients (the kernels need a GPU; the exchange plumbing does not).  A rank that issued another sequence
    would hang (the tests run this under a timeout); the summed gradient must come out identical on every rank."""
    import torch.distributed as dist
    backend, seen, equal, n_coll, overlapped = None, 1, None, 0, False
    cfg = CONFIGS[args.config]
    T = args.timesteps or cfg["T"]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SNN_DIST_BACKEND", "gloo"))
        backend = dist.get_backend()
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        from snn_for_object_detection_amd.trainer import FlatTrainer, broadcast_parameters, convert_sync_batchnorm
        classes = args.classes if args.classes is not None else cfg["classes"]
        model, lr = build_model(args, cfg, classes, "cpu")
        with torch.no_grad():                       # ranks start apart; the broadcast must bring them together
            for p in model.parameters():
                p.add_(0.01 * rank)
        trainer = FlatTrainer(model, lr=lr, find_unused_parameters=False)
        broadcast_parameters(trainer)
        n_coll += 1
        if args.sync_bn:
            convert_sync_batchnorm(model)
        bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d) and getattr(m, "_snn_sync_group", None)]
        overlapped = trainer._early_lo is not None
        for it in range(2):
            trainer.zero_grad()
            for m in bns:                           # forward: statistics of the global batch, one exchange per layer
                dist.all_reduce(torch.full((T, m.num_features, 2), float(rank), dtype=torch.float64),
                                group=m._snn_sync_group[0])
            g = torch.Generator().manual_seed(1000 * it + rank)
            trainer.flat_grad.copy_(torch.randn(trainer.flat_grad.shape, generator=g))
            for slot in trainer.slots:
                slot.written = True
            hook = getattr(model, "_snn_neck_grads_ready", None)
            if hook is not None:                    # the backward pass crosses the backbone / neck boundary
                hook()
            for m in reversed(bns):                 # backward: the BatchNorm-backward sums of the global batch
                dist.all_reduce(torch.full((T, m.num_features, 2), float(rank), dtype=torch.float64),
                                group=m._snn_sync_group[0])
            trainer._written_flags()
            trainer.all_reduce()
            n_coll = 1 + 2 * len(bns) + (2 if overlapped else 1)
        equal = replicas_equal(trainer, dist, "flat_grad") and replicas_equal(trainer, dist, "flat_param")
        dist.barrier()
    if rank == 0:
        emit_json({"metric": "event-frames/sec (BxT) SODa-3M GEN1 304x240 fwd+bwd", "value": None,
                   "unit": "event-frames/s", "n_gpus": world, "dry_run": True,
                   "config": {"name": args.config, "sync_batchnorm": bool(args.sync_bn and world > 1)},
                   "dist": {"backend": backend, "world_size": world, "ranks_in_allreduce": seen,
                            "replicas_equal_after_run": equal, "collectives_per_step": n_coll,
                            "overlapped_gradient_exchange": overlapped}})
    if world > 1:
        dist.destroy_process_group()
    return 0 if (equal is None or equal) else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="gen1",
                    help="workload: gen1 = BASELINE configs[1]/[2] (headline), 1mpx = configs[3], deep12 = configs[4]")
    ap.add_argument("--batch", type=int, default=None, help="samples per GPU (default: the config's)")
    ap.add_argument("--timesteps", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--classes", type=int, default=None)
    ap.add_argument("--sync-bn", action="store_true",
                    help="SyncBatchNorm as in the reference's config.yaml:76 (off by default, N>1 only)")
    ap.add_argument("--forward-precision", choices=("fp16x3", "bf16x6", "fp32", "bf16"), default="fp16x3",
                    help="forward conv arithmetic: fp16x3 = 2 fp16 pieces, 3 products (fp32-grade for |x| < 4094); "
                         "bf16x6 = 3 bf16 pieces, 6 products (fp32-grade, any range); fp32 = exact fp32 MFMA; "
                         "bf16 = the labelled THROUGHPUT mode (operands rounded to bf16, one product; not parity)")
    ap.add_argument("--backward-precision", choices=("bf16x3", "fp32", "bf16"), default="bf16x3",
                    help="arithmetic of the backward convolutions (bf16 = throughput mode, one product)")
    ap.add_argument("--storage", choices=("fp32", "bf16"), default="fp32",
                    help="activation storage: fp32 (default, the parity path) or bf16 - the labelled THROUGHPUT mode: wide "
                         "tensors bf16 in HBM, bf16 products, fp32 state / accumulation (own tolerances: "
                         "tests/test_gpu_bf16_storage.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="N=1 only: initialise the process group (RCCL unless SNN_DIST_BACKEND says otherwise) with ONE rank "
                         "and keep every collective of the N>1 step in the timed region - a rehearsal of the RCCL plumbing "
                         "on a one-GPU box, labelled as such in the JSON line")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launch / rendezvous / collective control flow only: no GPU work, no timing; "
                         "the JSON line carries n_gpus and dist but value null (CPU boxes, SNN_DIST_BACKEND=gloo)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves - BEFORE anything touches the GPU
        # (no HIP call has happened in this process; the ranks are child processes, this one only relays their exit code)
        raise SystemExit(spawn_ranks(args.gpus))
    cfg = CONFIGS[args.config]
    T = args.timesteps or cfg["T"]
    B = args.batch or cfg["batch"]
    H, W = args.height or cfg["H"], args.width or cfg["W"]
    classes = args.classes if args.classes is not None else cfg["classes"]
    big = args.config != "gen1"
    steps = args.steps if args.steps is not None else (6 if big else 20)
    warmup = args.warmup if args.warmup is not None else (2 if big else 5)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # never measure a different number of GPUs than the one asked for (the line would carry the wrong n_gpus)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    reserve_stdout()
    if args.dry_run:
        raise SystemExit(dry_run(args, world, rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if world > torch.cuda.device_count() and os.environ.get("SNN_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPUs (RCCL needs one GPU per "
                         "rank; SNN_DIST_BACKEND=gloo rehearses the control flow on fewer)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    distributed = world > 1 or args.rehearse_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # --rehearse-dist without a launcher
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL ("nccl") over xGMI is the production transport; SNN_DIST_BACKEND=gloo exists only to rehearse the
        # multi-process control flow on a box with fewer GPUs than ranks
        backend = os.environ.get("SNN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        backend = dist.get_backend()

    import snn_for_object_detection_amd as S
    from snn_for_object_detection_amd import _hip
    from snn_for_object_detection_amd.profiler import KernelProfiler
    from snn_for_object_detection_amd.trainer import FlatTrainer, broadcast_parameters
    _hip.load()
    S.functional.set_forward_precision(args.forward_precision)
    S.functional.set_backward_precision(args.backward_precision)
    S.functional.set_activation_storage(args.storage)
    sb = args.storage == "bf16"

    X, labels = synthetic_batch(T, B, H, W, classes, device, seed=rank, p=cfg["p"])  # a different shard per rank
    model, lr = build_model(args, cfg, classes, device)   # same reference init on every rank
    if args.config == "deep12":
        probe = deep12_probe(B, H, W, device)

        def loss_fn():
            # the loss reads the spikes of the last timestep only (as the detector does, models/soda.py:141-144)
            out, _ = model(X, last_only=True)
            return (out * probe).mean()
    else:
        def loss_fn():
            return model.training_step((X, labels))
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    # find_unused_parameters=False is what the reference's `strategy: ddp` means (config/config.yaml:35): every rank
    # produces every gradient, so the step carries no used-parameter flag exchange
    trainer = FlatTrainer(model, lr=lr, exchange_single_rank=args.rehearse_dist, find_unused_parameters=False)
    broadcast_parameters(trainer)
    if args.sync_bn and distributed:
        from snn_for_object_detection_amd.trainer import convert_sync_batchnorm
        convert_sync_batchnorm(model)

    def step():
        trainer.zero_grad()
        loss = loss_fn()
        loss.backward()
        trainer.step()
        return loss

    for _ in range(warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / steps
    frames_per_s = world * B * T * steps / elapsed
    replicas_eq = None
    if distributed:
        # evidence that the gradient exchange kept the replicas together: after `warmup + steps` updates every rank
        # holds the same weights (ranks saw different shards, so a missing or partial all-reduce would show here)
        replicas_eq = replicas_equal(trainer, dist)
    peak_gib = torch.cuda.max_memory_allocated() / 2**30

    roofline = None
    if not args.no_roofline:
        # every rank runs the two extra steps (they contain the all-reduce); only rank 0 brackets its launches
        # kernels are timed one at a time: the weight-gradient side stream is switched off for these two steps
        from snn_for_object_detection_amd import functional as HF
        prof = KernelProfiler() if rank == 0 else None
        _hip.PROFILER = prof
        side_stream_was, head_streams_was = HF.USE_WGRAD_STREAM, HF.USE_HEAD_STREAMS
        HF.USE_WGRAD_STREAM = HF.USE_HEAD_STREAMS = False
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        HF.USE_WGRAD_STREAM, HF.USE_HEAD_STREAMS = side_stream_was, head_streams_was
        _hip.PROFILER = None
    if rank == 0 and not args.no_roofline:
        table = prof.summary()
        total_ms = sum(r["ms"] for r in table.values())
        name, row = max(table.items(), key=lambda kv: kv[1]["ms"])
        default_shape = (T, B, H, W, classes) == (cfg["T"], cfg["batch"], cfg["H"], cfg["W"], cfg["classes"])
        traffic, traffic_src = (pmc_traffic(args.config + ("_bf16s" if sb else ""), name) if default_shape
                                else (None, {"note": "non-default shape"}))
        head = roofline_head(name, row, args.forward_precision, args.backward_precision, sb)
        roofline = {
            **head, "traffic": traffic, "traffic_source": traffic_src,
            "achieved_tflops": row["tflops"], "achieved_hbm_gbs": row["gbs"], "frac_hbm": row["gbs"] / PEAK_HBM_GBS,
            "dominant_template": dominant_template(prof), "avg_launch_us": row["avg_us"],
            "launches_per_step": row["calls"] // 2,
            "share_of_kernel_time": row["ms"] / total_ms,
            "flops_per_launch": row["flops"] / row["calls"],
            "all_kernels_ms_per_step": total_ms / 2,
            # every kernel family above 2 % of the kernel time against BOTH roofs (algorithmic work of its launches / their
            # time): the step is a long tail, no family holds a quarter of it
            "kernels": [
                {"kernel": k, "ms_per_step": r["ms"] / 2, "share": r["ms"] / total_ms, "launches_per_step": r["calls"] // 2,
                 "tflops": r["tflops"], "gbs": r["gbs"], "frac_hbm": r["gbs"] / PEAK_HBM_GBS,
                 "frac_mfma": (r["tflops"] / mfma_peak_for(k, args.forward_precision, args.backward_precision)[0]
                               if k.startswith("k_conv") else None),
                 "bound": roofline_head(k, r, args.forward_precision, args.backward_precision, sb)["bound"]}
                for k, r in sorted(table.items(), key=lambda kv: -kv[1]["ms"]) if r["ms"] / total_ms >= 0.02]
                + [c for c in [chain_row(table, prof.neuron_steps, 2, sb), wgrad_family_row(table, total_ms, 2)] if c],
            "timing": "HIP events around every C-ABI launch on its launch stream, weight-gradient side stream and head "
                      "streams off (every kernel alone on the GPU)",
        }
        if args.kernel_table:
            for k, r in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"[kernel] {k:48s} calls/step {r['calls'] // 2:4d}  avg {r['avg_us']:9.1f} us  "
                      f"{100 * r['ms'] / total_ms:5.1f}%  {r['tflops']:7.2f} TFLOP/s  {r['gbs']:8.1f} GB/s",
                      file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # every core this process may use, capped at 16 = one GPU's share of the box (the driver runs one bench per GPU;
        # `cores`, the host's total and the CPU model are all in the line)
        threads = min(host_cpu()["cores_available_to_process"] or 1, 16)
        cpu = cpu_baseline(args.config, H, W, classes, threads, cfg["p"])  # bounded sample: about 10-30 s of CPU work

    if distributed:
        dist.barrier()
    if rank == 0:
        exact = args.forward_precision == "fp32" and args.backward_precision == "fp32"
        out = {
            "metric": "event-frames/sec (BxT) SODa-3M GEN1 304x240 fwd+bwd" if args.config == "gen1"
                      else f"event-frames/sec (BxT) {cfg['label']} fwd+bwd",
            "value": frames_per_s,
            "unit": "event-frames/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": ms_per_step,
            # independent of n_gpus (weak scaling: every rank steps the same per-GPU batch): compare SCALE's N=1 line
            # with BENCH, and any N with any other, at a glance
            "per_gpu": {"event_frames_per_s": frames_per_s / world, "batch": B, "ms_per_step": ms_per_step},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # fp32 tensors and fp32 accumulation in every mode; "f32" alone only when the products are exact fp32 too
            "dtype": "bf16 storage + bf16 products, f32 state / accumulate (throughput mode, not parity)" if sb
                     else "f32" if exact else ("bf16 products, f32 accumulate / storage (throughput mode, not parity)"
                                          if "bf16" in (args.forward_precision, args.backward_precision)
                                          else "f32 (16-bit split products)"),
            "arithmetic_bits": {"storage": 16 if sb else 32, "accumulate": 32,
                                "forward_product": 8 if sb else PRODUCT_BITS[args.forward_precision],
                                "backward_product": 8 if sb else PRODUCT_BITS[args.backward_precision]},
            "data": "synthetic",
            "config": {
                "workload": f"{cfg['label']}, B={B}/GPU T={T}, p(event)={cfg['p']}, "
                            + ("2 boxes/sample, fwd + loss(last step)" if args.config != "deep12"
                               else "fwd + read-out loss(last step)")
                            + " + BPTT bwd + flat-grad all-reduce (N>1) + fused Adamax",
                "name": args.config, "trainable_params": n_params,
                "global_batch": B * world, "timesteps": T, "parallelism": f"dp{world}",
                "sync_batchnorm": bool(args.sync_bn and distributed),
                "storage": args.storage,
                "arithmetic": "bf16 STORAGE of the activation tensors (conv outputs, spikes, saved potentials, gradients), "
                              "fp32 neuron state / BatchNorm statistics / weights / accumulation; convolutions: stored bf16 "
                              "activations x bf16-rounded weights, one MFMA product" if sb else
                              "fp32 storage and accumulation; forward conv "
                              + {"fp16x3": "fp16x3 split products (two fp16 pieces per operand after exact 2^k "
                                           "pre-scaling, hh+hl+lh; fp32-grade: rel 5e-7 vs fp64, same as the fp32 MFMA)",
                                 "bf16x6": "bf16x6 split products (3-way bf16 split of both operands, fp32-grade: "
                                           "rel 5e-7 vs fp64, same as the fp32 MFMA)",
                                 "bf16": "bf16 single product (operands rounded to 8 significant bits: throughput mode)",
                                 "fp32": "exact fp32 MFMA"}[args.forward_precision]
                              + "; backward conv "
                              + {"bf16x3": "bf16x3 split products (hi*hi+hi*lo+lo*hi, rel 1e-5)",
                                 "bf16": "bf16 single product (throughput mode)",
                                 "fp32": "exact fp32 MFMA"}[args.backward_precision],
                "loss": float(loss.item()),
                "peak_hbm_gib": peak_gib,
                # what the opt-in bf16-STORAGE mode is held to (asserted on the GPU by tests/test_gpu_bf16_storage.py): not the
                # 1e-4 parity contract of the fp32 line
                **({"tolerances": {"convolution_rel_l2_vs_fp64_of_the_stored_operands": 3e-3,
                                   "weight_gradient_rel_l2_vs_fp64_of_the_stored_operands": 2e-5,
                                   "scans_and_batchnorm_apply": "the fp32 kernels' results rounded once to bf16 (bit-equal)",
                                   "training_step_loss_rel_vs_fp32_oracle": 0.06,
                                   "first_layer_weight_gradient_vs_fp32_oracle": "cosine > 0.6, rel. L2 < 1.0",
                                   "asserted_by": "tests/test_gpu_bf16_storage.py"}} if sb else {}),
            },
            "dist": {"backend": backend, "world_size": world,
                     "gradient_exchange": ("SUM all-reduce of the flat fp32 gradient: neck + head part started from a backward "
                                           "hook at the backbone / neck boundary (overlaps the backbone's backward "
                                           "pass), backbone part in step()" if getattr(trainer, "_early_lo", None)
                                           else "one SUM all-reduce of the flat fp32 gradient per step")
                     if distributed else None,
                     "replicas_equal_after_run": replicas_eq,
                     # (device, candidates probed, runs beside the main stream?) for the weight-gradient and the
                     # communication stream: HIP streams share a few hardware queues, see functional.concurrent_stream
                     "side_streams_probed": S.functional._STREAM_PROBE_LOG,
                     **({"rehearsal": "one-rank process group: every collective of the N>1 step is issued, none moves data "
                                      "between GPUs"} if args.rehearse_dist and world == 1 else {})},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if args.config == "gen1" and (H, W, classes) == (cfg["H"], cfg["W"], cfg["classes"]):
            # whole-step view against BASELINE.md section 2 (ideal-fusion byte model of this network: 114.0 M tensor elements
            # moved per event-frame fwd + bwd): 456.1 MB per frame with fp32 tensors, 228.0 MB with bf16 tensors, at 8 TB/s
            per_frame = 228.0e6 if sb else 456.1e6
            roof = PEAK_HBM_GBS * 1e9 / per_frame
            out["step_roofline"] = {"bound": "hbm", "ideal_bytes_per_frame": per_frame,
                                    "roof_frames_per_s_per_gpu": roof, "achieved_frames_per_s_per_gpu": frames_per_s / world,
                                    "frac": frames_per_s / world / roof,
                                    "basis": "BASELINE.md section 2: ideal-fusion bytes per event-frame ("
                                             + ("bf16" if sb else "fp32") + " tensors) at the 8 TB/s HBM3E spec"}
        emit_json(out)
    if distributed:
        dist.destroy_process_group()
    if replicas_eq is False:
        # an N > 1 line whose replicas drifted apart measured something else than data-parallel training: never exit 0
        raise SystemExit("bench.py: replicas differ after the run (replicas_equal_after_run false): the gradient exchange "
                         "did not keep the ranks together")


if __name__ == "__main__":
    main()
