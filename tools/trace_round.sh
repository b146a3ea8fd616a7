#!/bin/bash
# on the GPU box: kernel traces of the default step and of the opt-in head-streams step, reduced to the queue / gap report of
# tools/trace_gaps.py; + the host-side enqueue times of tools/sync_probe.py.   bash tools/trace_round.sh <tag>
out=gpurun_out/${1:-trace}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
B="--no-cpu-baseline --no-roofline --steps 6 --warmup 3"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t_default -o t -- python3 bench.py $B > $out/default.json 2> $out/default.err || exit 1
python tools/trace_gaps.py $(find $out/t_default -name 't_kernel_trace.csv') 4 > $out/trace_gaps_default.txt
export SNN_HEAD_STREAMS=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t_heads -o t -- python3 bench.py $B > $out/heads.json 2> $out/heads.err || exit 1
unset SNN_HEAD_STREAMS
python tools/trace_gaps.py $(find $out/t_heads -name 't_kernel_trace.csv') 4 > $out/trace_gaps_head_streams.txt
find $out -name '*kernel_trace.csv' -delete
python tools/sync_probe.py > $out/sync_probe_fp32.txt 2>&1
SNN_ACTIVATION_STORAGE=bf16 python tools/sync_probe.py > $out/sync_probe_bf16s.txt 2>&1
for f in trace_gaps_default trace_gaps_head_streams sync_probe_fp32 sync_probe_bf16s; do echo "== $f"; tail -9 $out/$f.txt; done
