#!/usr/bin/env python3
"""Matrix-pipe / VALU / LDS / texture-addresser utilisation per kernel family from rocprofv3 --pmc passes of the
bench workload (north_star: "evidenced by rocprof achieved-HBM-GB/s and MFMA utilisation").

    python tools/pmc_mfma.py out.json pass1/counter_collection.csv pass2/... [--time kernel_stats.csv]

Every pass is its own run of the same command (SQ has 8 counter slots, GRBM 2; gpurun refuses --pmc together with
the runtime trace domains).  Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles summed
over the SIMD / SQ instances; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Reported per family, averaged per launch:
  mfma_busy   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs)      matrix pipe busy share
  valu_share  = SQ_ACTIVE_INST_VALU x 4 / SQ_WAVE_CYCLES x 4 ... given as the share of wave-cycles issuing VALU
  wait_share  = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                              waves parked (waitcnt / barrier)
  lds_conflict= SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  ta_busy     = TA_BUSY_avr / (GRBM_GUI_ACTIVE/8)
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import family  # noqa: E402

NUM_CU, SIMD_PER_CU = 256, 4


def main():
    out_path, paths = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                fam = family(r["Kernel_Name"])
                agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[fam][r["Counter_Name"]] += 1
    out = {}
    for fam, d in agg.items():
        per = {c: v / max(cnt[fam][c], 1) for c, v in d.items()}   # per launch
        gui = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        wave = per.get("SQ_WAVE_CYCLES", 0.0)
        row = {"launches": max(cnt[fam].values()), "counters_per_launch": per}
        if gui > 0:
            row["gpu_cycles"] = gui
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
                row["mfma_busy"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * NUM_CU * SIMD_PER_CU)
            if "TA_BUSY_avr" in per:
                row["ta_busy"] = per["TA_BUSY_avr"] / gui
        if wave > 0:
            for key, c in (("valu_share", "SQ_ACTIVE_INST_VALU"), ("wait_share", "SQ_WAIT_ANY"),
                           ("issue_stall_share", "SQ_WAIT_INST_ANY"), ("lds_inst_share", "SQ_ACTIVE_INST_LDS")):
                if c in per:
                    row[key] = per[c] / wave
        if per.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0 and "SQ_LDS_BANK_CONFLICT" in per:
            row["lds_conflict"] = per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"]
        out[fam] = row
    from bench import csrc_fingerprint  # noqa: E402  (repo root is on sys.path through pmc_traffic's import)
    out["_meta"] = {"csrc_sha256": csrc_fingerprint(), "passes": [os.path.basename(os.path.dirname(p)) for p in paths]}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    rows = [(v.get("gpu_cycles", 0.0) * v["launches"], k, v) for k, v in out.items() if k != "_meta"]
    tot = sum(r[0] for r in rows) or 1.0
    print(f"{'kernel family':50s} {'share':>6s} {'MFMA':>6s} {'VALU':>6s} {'wait':>6s} {'LDScf':>6s} {'TA':>6s}")
    for g, k, v in sorted(rows, reverse=True)[:20]:
        def pct(key):
            return f"{100 * v[key]:5.1f}%" if key in v else "    - "
        print(f"{k[:50]:50s} {100 * g / tot:5.1f}% {pct('mfma_busy')} {pct('valu_share')} {pct('wait_share')} "
              f"{pct('lds_conflict')} {pct('ta_busy')}")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    main()
