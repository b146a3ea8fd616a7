#!/bin/bash
# on the GPU box: the profile round of the three other workloads, one after the other (each leaves gpurun_out/<tag>_<cfg>/)
# usage: bash tools/profile_all.sh <tag> [configs...]      (default: deep12 1mpx gen1:bf16)
tag=${1:-p}; shift
cfgs=${@:-"deep12 1mpx gen1:bf16"}
for c in $cfgs; do
  cfg=${c%%:*}; mode=""
  [ "$c" != "$cfg" ] && mode=${c##*:}
  sfx=""; [ -n "$mode" ] && sfx="_$mode"
  echo "== $cfg $mode" 
  ROUND=${ROUND:-r04} bash tools/profile_round.sh ${tag}_$cfg$sfx $cfg $mode > gpurun_out/${tag}_$cfg$sfx.log 2>&1 || { echo "FAILED $cfg $mode"; tail -5 gpurun_out/${tag}_$cfg$sfx.log; exit 1; }
  tail -2 gpurun_out/${tag}_$cfg$sfx/traffic.txt
done
