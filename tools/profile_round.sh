#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/
#   bench.json + kernel_table.txt : python bench.py --kernel-table (HIP-event kernel table, roofline, CPU baseline)
#   stats/s_kernel_stats.csv      : rocprofv3 --kernel-trace --stats of a 5-step bench run
#   traffic.json                  : per-kernel HBM bytes from two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE)
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
python bench.py --kernel-table > $out/bench.json 2> $out/kernel_table.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- \
    python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -o f -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $out/f.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -o w -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2> $out/w.err || exit 1
python tools/pmc_traffic.py $(find $out/pmc_f -name 'f_counter_collection.csv') $(find $out/pmc_w -name 'w_counter_collection.csv') $out/traffic.json
find $out -name '*kernel_trace.csv' -delete
ls -R $out | head -40
