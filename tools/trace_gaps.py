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

# ---- gaps of the busiest queue (the main stream): size classes and the largest ones with their neighbours
main_q = max(per_q, key=per_q.get)
mq = [e for e in ev if e[2] == main_q]
gaps = [(mq[k + 1][0] - mq[k][1], mq[k][3], mq[k + 1][3]) for k in range(len(mq) - 1)]
classes = [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)]
print(f"gaps between consecutive kernels of queue {main_q} (per step):")
for lo, hi in classes:
    sel = [g for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
    print(f"  {lo:3.0f}-{hi if hi < 1e9 else float('inf'):4.0f} us: {len(sel) / steps:6.1f} gaps, {sum(g[0] for g in sel) / 1e6 / steps:6.3f} ms")
neg = [g for g in gaps if g[0] < 0]
print(f"  overlapping (next starts before the previous ends): {len(neg) / steps:.1f}")


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n[:70]


print("largest gaps:")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  {g / 1e3:8.1f} us   after {short(a)}   before {short(b)}")
# how much of a main-queue gap is covered by a kernel of another queue
other = sorted((s, e) for s, e, q, _ in ev if q != main_q)
covered = 0
j = 0
for k in range(len(mq) - 1):
    gs, ge = mq[k][1], mq[k + 1][0]
    if ge <= gs:
        continue
    for s, e in other:
        if e <= gs:
            continue
        if s >= ge:
            break
        covered += min(e, ge) - max(s, gs)
tot_gap = sum(g[0] for g in gaps if g[0] > 0)
print(f"main-queue gaps: {tot_gap / 1e6 / steps:.3f} ms per step, of which {covered / 1e6 / steps:.3f} ms have a kernel of another queue running")
