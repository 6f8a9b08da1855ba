#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <shape> <fwd|dgrad|wgrad>   (environment selects the precision mode / tuning variables)
# Collects SQ counters for one conv kernel in separate rocprofv3 --pmc passes and prints the per-dispatch averages.
tag=$1; shape=$2; what=$3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_${tag}_$i -o p --output-format csv -- python3 $R/tools/one_conv.py $shape $what > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${tag}_$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$R/gpurun_out/pmc_${tag}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_igemm' not in k and 'conv_wgrad' not in k: continue
        a = agg[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('$tag', {k: round(v[0] / v[1]) for k, v in sorted(agg.items())})
PY
