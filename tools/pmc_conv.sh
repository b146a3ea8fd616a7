#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <one_conv args...>   -- SQ counter passes for one conv entry point
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$PMC_GROUPS" ]; then IFS=";" read -ra GROUPS_ <<< "$PMC_GROUPS"; else GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU GRBM_GUI_ACTIVE"); fi
for grp in "${GROUPS_[@]}"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmcc_$tag/$n -o c -- python3 tools/one_conv.py "$@" > gpurun_out/pmcc_$tag.log 2>&1 || echo "pass failed: $grp"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmcc_$tag/*/c_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'k_conv' not in k: continue
        agg[k[:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf gpurun_out/pmcc_$tag/*/c_kernel_trace.csv
