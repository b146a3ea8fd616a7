#!/bin/bash
# usage (GPU box): bash tools/wb_prof.sh <tag>  -> per-kernel average durations of tools/wgrad_bench.py (rocprofv3 kernel trace)
tag=${1:-wb}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/trace -o w -- python3 tools/wgrad_bench.py > gpurun_out/$tag/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/$tag/trace/**/w_kernel_trace.csv",recursive=True)[0]
d=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "wgrad" not in r["Kernel_Name"]: continue
    k=(r["Kernel_Name"].replace("void (anonymous namespace)::","")[:34], r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""))
    d.setdefault(k,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print(f"{k[0]:36s} grid {k[1]:>8s} x {k[2]:>3s}  n={len(v):3d}  avg {sum(v)/len(v):8.1f} us")
PY
grep -v amdgpu gpurun_out/$tag/bench.log | tail -11
rm -rf gpurun_out/$tag/trace
