#!/bin/bash
# Round profile of the default bench command on the GPU box: kernel trace + stats, then the two PMC passes for HBM traffic.
# usage (inside gpurun): tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*
tag=${1:-round2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# the default step: hipGraph replay, linear (one kernel at a time), so per-kernel durations are exclusive
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_stats -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/${tag}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 3 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/${tag}_pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 3 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/${tag}_pmc_write.log 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_pmc_fetch $R/gpurun_out/${tag}_pmc_write $R/gpurun_out/${tag}_pmc_traffic.json > /dev/null
# the counter CSVs are large: keep only the aggregate
rm -rf $R/gpurun_out/${tag}_pmc_fetch $R/gpurun_out/${tag}_pmc_write
rm -f $R/gpurun_out/${tag}_stats/p_kernel_trace.csv
echo ok
