#!/bin/bash
# rocprofv3 kernel stats of the graph-replayed step in one conv arithmetic: tools/r3_prof.sh <tag> <mode>
tag=$1; mode=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DSRL_CONV_PRECISION=$mode
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
rm -f $R/gpurun_out/${tag}_stats/p_kernel_trace.csv
python3 $R/tools/kstats.py $R/gpurun_out/${tag}_stats/p_kernel_stats.csv 14 25 > $R/gpurun_out/${tag}_kstats.txt
cat $R/gpurun_out/${tag}_kstats.txt
