#!/bin/bash
# round-3 record: the default bench line, then the rocprofv3 kernel stats + PMC traffic of the same step (tools/profile_round.sh)
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r3_bench.json
bash tools/profile_round.sh r3 && python3 tools/kstats.py gpurun_out/r3_stats/p_kernel_stats.csv 14 > gpurun_out/r3_kstats.txt; head -3 gpurun_out/r3_kstats.txt
