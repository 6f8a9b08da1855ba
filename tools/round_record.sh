#!/bin/bash
# round record: the default bench line, then the rocprofv3 kernel stats + PMC traffic of the same step (tools/profile_round.sh <tag>).  usage: tools/round_record.sh [tag, default r4]
tag=${1:-r4}
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/${tag}_bench.json
bash tools/profile_round.sh ${tag} && python3 tools/kstats.py gpurun_out/${tag}_stats/p_kernel_stats.csv 14 > gpurun_out/${tag}_kstats.txt; head -3 gpurun_out/${tag}_kstats.txt
