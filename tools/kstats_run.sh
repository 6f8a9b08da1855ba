#!/bin/bash
# kernel stats of the default bench step under an environment setting: tools/kstats_run.sh <tag> [VAR=value ...]  -> gpurun_out/<tag>_kstats.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/${tag}_stats/p_kernel_stats.csv 14 40 > $R/gpurun_out/${tag}_kstats.txt
rm -f $R/gpurun_out/${tag}_stats/p_kernel_trace.csv
