#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench workload.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv out.json

``out.json`` also gets a ``_meta`` entry with the sha256 of the kernel sources the passes ran on: ``bench.py`` reports
``roofline.traffic`` from the file only while that fingerprint matches the current sources.

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced reads, so the read side is DOUBLED; WRITE_SIZE is exact.
Kernels are grouped by template family (text before '<' / '('), averages are per launch.
"""
import collections
import csv
import json
import re
import sys


BF16S = False   # argv[5] == "bf16s": the passes ran bench.py --storage bf16; labels get profiler.py's ", bf16s" suffix


def family(name: str) -> str:
    fam = _family(name)
    if BF16S and (fam.startswith(("k_conv_gather", "k_conv_wgrad", "k_conv_s2dgrad3", "k_conv_first", "k_affine_neuron",
                                  "k_bn_stats", "k_bn_bwd_apply"))):
        return fam + ", bf16s"
    return fam


def _family(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+)", name)
    base = m.group(1) if m else name
    if base == "k_conv_gather":  # <BN, WM, WN, DGRAD, VEC, ...>: the first five arguments, as profiler.py labels them
        t = re.search(r"k_conv_gather<([^>]*)>", name)
        return f"k_conv_gather<{', '.join(a.strip() for a in t.group(1).split(',')[:5])}>" if t else base
    if base == "k_conv_halo3":  # <CO, F16, ABL, BNAP>: forward (fp16 pieces) or data gradient, as profiler.py labels them
        t = re.search(r"k_conv_halo3<(\d+), (true|false)", name)
        if t and BF16S:
            return f"k_conv_halo3<{t.group(1)}, bf16s>"   # one instance serves forward and data gradient
        return f"k_conv_halo3<{t.group(1)}, {'fwd' if t.group(2) == 'true' else 'dgrad'}>" if t else base
    if base == "k_conv_s2dgrad3":
        return "k_conv_s2dgrad3<dgrad>"
    if base == "k_conv_first":  # <CIN, KS, WGRAD>: the weight-gradient instance belongs to snn_conv2d_wgrad
        return "k_conv_wgrad" if re.search(r"k_conv_first<[^>]*true>", name) else "k_conv_first<2, 3, false>"
    if base in ("k_affine_neuron_fwd", "k_affine_neuron_bwd"):   # <NEURON, ...>: profiler.py labels the neuron
        t = re.search(base + r"<(\d+)", name)
        return f"{base}<{t.group(1)}>" if t else base
    if base.startswith("k_conv_wgrad") or base == "k_wgrad_reduce":
        return "k_conv_wgrad"  # one snn_conv2d_wgrad call = one tile kernel (any variant) + its ordered reduce
    return base


def load(path, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = family(r["Kernel_Name"])
            per[k][0] += float(r["Counter_Value"])
            per[k][1] += 0 if "k_wgrad_reduce" in r["Kernel_Name"] else 1  # the reduce belongs to its tile kernel's call
    return per


def main():
    global BF16S
    BF16S = len(sys.argv) > 5 and sys.argv[5] == "bf16s"
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        fs, fn = fetch.get(k, [0.0, 0])
        ws, wn = write.get(k, [0.0, 0])
        rd = 2.0 * 1024.0 * fs / max(fn, 1)   # gfx950 correction: x2
        wr = 1024.0 * ws / max(wn, 1)
        out[k] = {"launches": max(fn, wn), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_fingerprint
    rows = dict(out)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # steps the profiled command ran (warm-up included)
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows.values())
    out["_meta"] = {"csrc_sha256": csrc_fingerprint(), "units": "bytes per launch; FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024",
                    "passes": [os.path.basename(os.path.dirname(a)) for a in sys.argv[1:3]],
                    "profiled_steps": steps, "hbm_bytes_all_kernels": total,
                    "hbm_bytes_per_step": total / steps if steps else None}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:16]:
        print(f"{k:52s} n={v['launches']:5d}  read {v['read_bytes_per_launch'] / 1e6:9.1f} MB  "
              f"write {v['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
