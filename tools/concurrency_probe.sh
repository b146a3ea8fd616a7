#!/bin/bash
# How much would finer-grained overlap of matrix-bound and HBM-bound kernels buy?  Upper-bound probe: two independent
# half-length steps (T=16) in two PROCESSES sharing the GPU against one of them alone.  bash tools/concurrency_probe.sh <tag>
out=gpurun_out/${1:-conc}
mkdir -p $out
B="--no-cpu-baseline --no-roofline --steps 40 --warmup 8"
python bench.py $B --timesteps 16 > $out/half_alone.json 2>/dev/null || exit 1
python bench.py $B --timesteps 32 > $out/full_alone.json 2>/dev/null || exit 1
python bench.py $B --timesteps 16 > $out/half_a.json 2>/dev/null &
pa=$!
python bench.py $B --timesteps 16 > $out/half_b.json 2>/dev/null &
pb=$!
wait $pa || exit 1
wait $pb || exit 1
for f in half_alone full_alone half_a half_b; do
  python -c "import json;d=json.load(open('$out/$f.json'));print('$f', round(d['ms_per_step'],3), 'ms/step', round(d['value']), 'frames/s')" | tee -a $out/summary.txt
done
