#!/bin/bash
# usage (in the container, after gpurun merged the outputs): bash tools/collect_profiles.sh gpurun_out/<tag> <config> [round]
# copies what tools/profile_round.sh wrote into profiles/ under the names DESIGN.md / bench.py refer to
src=$1; cfg=$2; r=${3:-r04}
set -e
cp $src/bench.json profiles/${r}_bench_${cfg}.json
cp $src/kernel_table.txt profiles/${r}_bench_${cfg}_kernel_table.txt
cp $src/stats1/s_kernel_stats.csv profiles/${r}_bench_${cfg}_kernel_stats_single_stream.csv
cp $src/stats2/s_kernel_stats.csv profiles/${r}_bench_${cfg}_kernel_stats_two_streams.csv
cp $src/traffic.json profiles/${r}_pmc_traffic_${cfg}.json
cp $src/traffic_counters.csv profiles/${r}_pmc_counters_${cfg}.csv
cp $src/mfma.json profiles/${r}_pmc_mfma_${cfg}.json
cp $src/mfma.txt profiles/${r}_pmc_mfma_${cfg}.txt
ls -la profiles/${r}_*${cfg}*
