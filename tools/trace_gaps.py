#!/usr/bin/env python3
"""Analyse a rocprofv3 kernel trace: busy time per queue, idle gaps, overlap between queues.
usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ev.sort()
# steady state: the window of the last `steps` optimiser kernels (one k_adamax per step)
marks = [e[0] for e in ev if "k_adamax" in e[3]]
if len(marks) > steps:
    ev = [e for e in ev if marks[-steps - 1] <= e[0] < marks[-1]]
span = ev[-1][1] - ev[0][0]
per_q = defaultdict(int)
for s, e, q, _ in ev:
    per_q[q] += e - s
print(f"window {span / 1e6:.2f} ms = {steps} steps, kernels {len(ev)}")
for q, b in per_q.items():
    print(f"  queue {q}: busy {b / 1e6:.2f} ms ({100 * b / span:.1f} %)")
# union busy time over all queues
cur_s, cur_e, union = None, None, 0
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"  any kernel running: {union / 1e6:.2f} ms ({100 * union / span:.1f} %), idle {100 - 100 * union / span:.1f} %")
