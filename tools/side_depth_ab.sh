#!/bin/bash
# same-call A/B of the weight-gradient side stream's allowed lag (functional.WGRAD_SIDE_DEPTH): bash tools/side_depth_ab.sh <tag>
out=gpurun_out/${1:-depth}
mkdir -p $out
for d in 1 2 4 8 1 4; do
  SNN_WGRAD_SIDE_DEPTH=$d python bench.py --no-cpu-baseline --no-roofline > $out/d$d.json 2>/dev/null || exit 1
  python -c "import json;d=json.load(open('$out/d$d.json'));print('depth $d', round(d['ms_per_step'],3), round(d['config']['peak_hbm_gib'],1))" | tee -a $out/summary.txt
done
