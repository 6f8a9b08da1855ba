#!/bin/bash
# usage: tools/pmc_mem.sh <tag> <shape> <fwd|dgrad|wgrad>: memory-pipeline counters (TA / TCP / TCC) of one conv kernel, separate --pmc passes
tag=$1; shape=$2; what=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcm_${tag}_$i -o p --output-format csv -- python3 $R/tools/one_conv.py $shape $what > $R/gpurun_out/pmcm_${tag}_$i.log 2>&1 || { tail -3 $R/gpurun_out/pmcm_${tag}_$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$R/gpurun_out/pmcm_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_igemm' not in k and 'conv_wgrad' not in k: continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('$tag', {k: round(v[0] / v[1]) for k, v in sorted(agg.items())})
PY
rm -rf $R/gpurun_out/pmcm_${tag}_*
