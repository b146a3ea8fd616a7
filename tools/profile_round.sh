#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag> [config] [bf16]     -> gpurun_out/<tag>/
#   third argument "bf16": the same passes on the bf16-STORAGE throughput mode (bench.py --storage bf16)
#   bench.json + kernel_table.txt : python bench.py --kernel-table (HIP-event kernel table, roofline, CPU baseline)
#   stats1/s_kernel_stats.csv     : rocprofv3 --kernel-trace --stats, weight-gradient side stream OFF (every kernel alone
#                                   on the GPU: the per-kernel averages behind roofline.frac)
#   stats2/s_kernel_stats.csv     : the same with the side stream ON (the step as it really runs)
#   traffic.json                  : per-kernel HBM bytes from two SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE)
#   traffic_counters.csv          : the same two passes summed per kernel NAME (what traffic.json is recomputed from)
#   mfma.json                     : matrix pipe / VALU / LDS / TA utilisation per kernel family (three more --pmc passes)
tag=${1:-prof}
cfg=${2:-gen1}
ROUND=${ROUND:-r04}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
mode=${3:-fp32}
sfx=""
B="--config $cfg --no-cpu-baseline"
if [ "$mode" = "bf16" ]; then B="$B --storage bf16"; sfx="_bf16s"; fi
export SNN_NO_WGRAD_STREAM=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o s -- \
    python3 bench.py $B --steps 5 --warmup 2 --no-roofline > $out/stats1_bench.json 2> $out/stats1.err || exit 1
unset SNN_NO_WGRAD_STREAM
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats2 -o s -- \
    python3 bench.py $B --steps 5 --warmup 2 --no-roofline > $out/stats2_bench.json 2> $out/stats2.err || exit 1
export SNN_NO_WGRAD_STREAM=1
P="$B --steps 2 --warmup 1 --no-roofline"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -o f -- \
    python3 bench.py $P > /dev/null 2> $out/f.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -o w -- \
    python3 bench.py $P > /dev/null 2> $out/w.err || exit 1
python tools/pmc_traffic.py $(find $out/pmc_f -name 'f_counter_collection.csv') $(find $out/pmc_w -name 'w_counter_collection.csv') $out/traffic.json 3 ${sfx#_} $out/traffic_counters.csv > $out/traffic.txt
[ -z "$sfx" ] && sed -i 's/"counters": "traffic_counters.csv"/"counters": "'$ROUND'_pmc_counters_'$cfg'.csv"/' $out/traffic.json
[ -n "$sfx" ] && sed -i 's/"counters": "traffic_counters.csv"/"counters": "'$ROUND'_pmc_counters_'$cfg$sfx'.csv"/' $out/traffic.json
# the bench line of this round quotes THESE passes: put the file where bench.py looks for it (tools/collect_profiles.sh
# copies the same file to the same place in the repository afterwards)
cp $out/traffic.json profiles/${ROUND}_pmc_traffic_$cfg$sfx.json
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $out/pmc_a -o a -- python3 bench.py $P > /dev/null 2> $out/a.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d $out/pmc_b -o b -- python3 bench.py $P > /dev/null 2> $out/b.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE \
    --output-format csv -d $out/pmc_c -o c -- python3 bench.py $P > /dev/null 2> $out/c.err || exit 1
python tools/pmc_mfma.py $out/mfma.json $(find $out/pmc_a -name 'a_counter_collection.csv') \
    $(find $out/pmc_b -name 'b_counter_collection.csv') $(find $out/pmc_c -name 'c_counter_collection.csv') > $out/mfma.txt
unset SNN_NO_WGRAD_STREAM
python bench.py ${B/--no-cpu-baseline/} --kernel-table > $out/bench.json 2> $out/kernel_table.txt || exit 1
find $out -name '*kernel_trace.csv' -delete
find $out -name '*counter_collection.csv' -delete
ls -R $out | head -60
