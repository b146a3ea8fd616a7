#!/usr/bin/env python3
"""Per-kernel texture-addresser load from one rocprofv3 --pmc pass (TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum
GRBM_GUI_ACTIVE) of the bench workload: which kernels are bound by the NUMBER of cache accesses rather than bytes.
usage: pmc_ta.py counter_collection.csv"""
import collections
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import family  # noqa: E402

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    f = family(r["Kernel_Name"])
    agg[f][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[f] += 1
rows = []
for f, d in agg.items():
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # summed over the 8 XCDs
    if gui <= 0:
        continue
    rows.append((gui, f, d.get("TA_BUSY_avr", 0.0) / gui, d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / 256.0 / gui, cnt[f]))
tot = sum(r[0] for r in rows)
print(f"{'kernel family':52s} {'share':>6s} {'TA busy':>8s} {'L1 acc/clk/CU':>14s} {'launches':>8s}")
for gui, f, ta, acc, n in sorted(rows, reverse=True)[:24]:
    print(f"{f[:52]:52s} {100 * gui / tot:5.1f}% {100 * ta:7.1f}% {acc:14.3f} {n:8d}")
