#!/bin/bash
# same-call A/B of an environment knob: bash tools/ab.sh <tag> <ENV=VALUE> [rounds] [extra bench flags...]
# runs bench.py alternately without / with the knob and prints ms/step of every run (boxes differ by +-2 %: compare inside ONE call)
tag=$1; knob=$2; rounds=${3:-2}; shift 3
out=gpurun_out/$tag
mkdir -p $out
for r in $(seq 1 $rounds); do
  python bench.py --no-cpu-baseline --no-roofline "$@" > $out/base_$r.json 2>$out/base_$r.err || { tail -5 $out/base_$r.err; exit 1; }
  env $knob python bench.py --no-cpu-baseline --no-roofline "$@" > $out/knob_$r.json 2>$out/knob_$r.err || { tail -5 $out/knob_$r.err; exit 1; }
  python - <<PY | tee -a $out/summary.txt
import json
b=json.load(open('$out/base_$r.json')); k=json.load(open('$out/knob_$r.json'))
print('round $r: default', round(b['ms_per_step'],3), ' $knob', round(k['ms_per_step'],3), ' loss', b['config']['loss'], k['config']['loss'])
PY
done
