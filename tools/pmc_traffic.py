#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench workload.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv \
        out.json <profiled steps> [bf16s] [counters.csv]

Two files come out:
* ``counters.csv`` (sixth argument; default ``out.json`` with ``_counters.csv``): the raw passes summed PER KERNEL NAME
  (name, counter, launches, sum in KiB).  The raw ``*_counter_collection.csv`` of a step are hundreds of MB; this is the
  compact form that is committed under ``profiles/`` so that every figure below can be recomputed from a committed file
  (``python tools/pmc_traffic.py --from-counters profiles/rNN_pmc_counters_<config>.csv out.json <steps> [bf16s]``;
  ``tests/test_host_logic.py`` does exactly that).
* ``out.json``: bytes per launch per kernel FAMILY, plus ``_meta`` with the sha256 of the kernel sources the passes ran on
  (``bench.py`` reports ``roofline.traffic`` from the file only while that fingerprint matches the current sources) and
  the whole-step total.

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so the read side is DOUBLED; WRITE_SIZE is exact.

Families.  A family is what ONE C-ABI call launches: ``family(name)`` names it, ``counts_as_launch(name)`` says whether
the kernel is the call's main kernel (its launches are the family's launches) or a helper whose BYTES belong to the
family while its launches do not (``HELPERS``: the ordered split-K reduce behind every weight-gradient tile kernel).
A kernel that is neither is its own family; ``tests/test_host_logic.py`` checks on the committed rocprofv3 ``--stats``
csv that no family ends up with bytes and zero launches (round 3 dropped ``k_wgrad_reduce4``'s 1.55 GB per step that way).
"""
import collections
import csv
import json
import os
import re
import sys


BF16S = False   # "bf16s": the passes ran bench.py --storage bf16; labels get profiler.py's ", bf16s" suffix

# helper kernels: prefix of the kernel's base name -> the family whose calls launch them.  The ordered split-K reduce runs
# behind all three weight-gradient kernels (different programs with different roofs: round 4 prices them separately); the
# raw pass tags every reduce launch with the kernel it followed in dispatch order ("... [after k_conv_wgrad_halo]"), an
# untagged one (counter files of before) counts for the implicit GEMM.
HELPERS = {"k_wgrad_reduce": "k_conv_wgrad_pipe"}
WGRAD_FAMILIES = ("k_conv_wgrad_pipe", "k_conv_wgrad_halo", "k_conv_first<wgrad>")


def base_name(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+)", name)
    return m.group(1) if m else name


def counts_as_launch(name: str) -> bool:
    """False for helper kernels: their bytes are added to their family, their launches are not."""
    return not any(base_name(name).startswith(h) for h in HELPERS)


def family(name: str, bf16s: bool = None) -> str:
    bf16s = BF16S if bf16s is None else bf16s
    fam = _family(name, bf16s)
    if bf16s and (fam.startswith(("k_conv_gather", "k_conv_wgrad", "k_conv_s2dgrad3", "k_conv_first", "k_affine_neuron",
                                  "k_bn_stats", "k_bn_bwd_apply"))):
        return fam + ", bf16s"
    return fam


def _family(name: str, bf16s: bool) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    base = base_name(name)
    tagged = re.search(r"\[after ([^\]]+)\]", name)
    for helper, fam in HELPERS.items():
        if base.startswith(helper):
            return tagged.group(1) if tagged else fam
    if base == "k_conv_gather":  # <BN, WM, WN, DGRAD, VEC, ...>: the first five arguments, as profiler.py labels them
        t = re.search(r"k_conv_gather<([^>]*)>", name)
        return f"k_conv_gather<{', '.join(a.strip() for a in t.group(1).split(',')[:5])}>" if t else base
    if base == "k_conv_halo3":  # <CO, F16, ABL, BNAP>: forward (fp16 pieces) or data gradient, as profiler.py labels them
        t = re.search(r"k_conv_halo3<(\d+), (true|false)", name)
        if t and bf16s:
            return f"k_conv_halo3<{t.group(1)}, bf16s>"   # one instance serves forward and data gradient
        return f"k_conv_halo3<{t.group(1)}, {'fwd' if t.group(2) == 'true' else 'dgrad'}>" if t else base
    if base == "k_conv_s2dgrad3":
        return "k_conv_s2dgrad3<dgrad>"
    if base == "k_conv_first":  # <CIN, KS, WGRAD, BNAPPLY, SB>: the weight-gradient instances belong to snn_conv2d_wgrad
        t = re.search(r"k_conv_first<([^>]*)>", name)
        args = [a.strip() for a in t.group(1).split(",")] if t else []
        return "k_conv_first<wgrad>" if len(args) > 2 and args[2] == "true" else "k_conv_first<2, 3, false>"
    if base in ("k_affine_neuron_fwd", "k_affine_neuron_bwd"):   # <NEURON, ...>: profiler.py labels the neuron
        t = re.search(base + r"<(\d+)", name)
        return f"{base}<{t.group(1)}>" if t else base
    if base == "k_conv_wgrad_halo":
        return "k_conv_wgrad_halo"   # the halo-resident 3x3 weight gradient (csrc/wgrad_halo.hip)
    if base.startswith("k_conv_wgrad"):
        return "k_conv_wgrad_pipe"   # the implicit GEMM, pipelined or exact fp32: every tile variant (+ its ordered reduce)
    return base


def aggregate(path, counter):
    """raw rocprofv3 counter csv -> {kernel name: [sum of the counter (KiB), launches]}"""
    per = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        rows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
    if rows and "Dispatch_Id" in rows[0]:
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))   # program order (the passes run on ONE stream)
    last_wgrad = None
    for r in rows:
        name = r["Kernel_Name"]
        if counts_as_launch(name):
            fam = _family(name, False)
            if fam in WGRAD_FAMILIES:
                last_wgrad = fam
        elif last_wgrad is not None:
            name = f"{name} [after {last_wgrad}]"   # a helper belongs to the call whose main kernel it followed
        row = per[name]
        row[0] += float(r["Counter_Value"])
        row[1] += 1
    return per


def write_counters(path, fetch, write):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Launches", "Sum_KiB"])
        for counter, per in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for name in sorted(per):
                w.writerow([name, counter, per[name][1], repr(per[name][0])])


def read_counters(path):
    fetch, write = {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            (fetch if r["Counter_Name"] == "FETCH_SIZE" else write)[r["Kernel_Name"]] = [float(r["Sum_KiB"]),
                                                                                           int(r["Launches"])]
    return fetch, write


def by_family(per, bf16s=None):
    """{kernel name: [KiB, launches]} -> {family: [KiB, launches of the family's main kernels]}"""
    out = collections.defaultdict(lambda: [0.0, 0])
    for name, (kib, n) in per.items():
        row = out[family(name, bf16s)]
        row[0] += kib
        row[1] += n if counts_as_launch(name) else 0
    return out


def traffic_table(fetch, write, steps, bf16s=None):
    """Per-family bytes per launch + the whole-step total; raises when a family has bytes but no launch to charge them to."""
    fetch, write = by_family(fetch, bf16s), by_family(write, bf16s)
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(k, [0.0, 0])
        ws, wn = write.get(k, [0.0, 0])
        n = max(fn, wn)
        if n == 0:
            raise RuntimeError(f"pmc_traffic: family {k!r} has counter values but no launch-counting kernel (a helper "
                               "kernel whose main kernel never ran, or a missing HELPERS rule)")
        rd = 2.0 * 1024.0 * fs / n   # gfx950 correction: x2
        wr = 1024.0 * ws / n
        out[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
    # the Norm + neuron CHAIN (5 tensors per neuron-timestep): forward scan + reverse scan + BatchNorm-backward
    # apply, as one figure
    chain = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in out.items()
                if k.startswith(("k_affine_neuron", "k_bn_bwd_apply")))
    meta = {"units": "bytes per launch; FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024", "profiled_steps": steps,
            "hbm_bytes_all_kernels": total, "hbm_bytes_per_step": total / steps if steps else None,
            "norm_neuron_chain_bytes_per_step": chain / steps if steps else None,
            "helpers": {h: f"bytes added to {fam}, launches not counted" for h, fam in HELPERS.items()}}
    return out, meta


def main():
    global BF16S
    argv = sys.argv[1:]
    from_counters = argv and argv[0] == "--from-counters"
    if from_counters:
        counters_path, out_path = argv[1], argv[2]
        rest = argv[3:]
        fetch, write = read_counters(counters_path)
        passes = [os.path.basename(counters_path)]
    else:
        fetch, write = aggregate(argv[0], "FETCH_SIZE"), aggregate(argv[1], "WRITE_SIZE")
        out_path = argv[2]
        rest = argv[3:]
        passes = [os.path.basename(os.path.dirname(a)) for a in argv[:2]]
    steps = int(rest[0]) if rest else 0   # steps the profiled command ran (warm-up included)
    BF16S = len(rest) > 1 and rest[1] == "bf16s"
    if not from_counters:
        counters_path = rest[2] if len(rest) > 2 else out_path.replace(".json", "") + "_counters.csv"
        write_counters(counters_path, fetch, write)
    rows, meta = traffic_table(fetch, write, steps)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_fingerprint
    out = dict(rows)
    out["_meta"] = {"csrc_sha256": csrc_fingerprint(), "passes": passes, "counters": os.path.basename(counters_path),
                    "bf16s": BF16S, **meta}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:16]:
        print(f"{k:52s} n={v['launches']:5d}  read {v['read_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"write {v['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")
    print(f"whole step: {meta['hbm_bytes_per_step'] / 1e9 if meta['hbm_bytes_per_step'] else float('nan'):.2f} GB; "
          f"Norm + neuron chain: "
          f"{meta['norm_neuron_chain_bytes_per_step'] / 1e9 if meta['norm_neuron_chain_bytes_per_step'] else float('nan'):.2f} GB")


if __name__ == "__main__":
    main()
