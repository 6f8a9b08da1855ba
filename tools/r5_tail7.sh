#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/t7_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t7_tests.txt
for v in 1 0 1 0; do
  DSRL_CONVT_CE=$v timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-prof --no-cpu-baseline --no-config5 > gpurun_out/t4_bench_$v.json 2> gpurun_out/t4_bench_$v.err || exit 1
  python - <<EOF
import json
d=json.loads(open('gpurun_out/t4_bench_$v.json').read().strip().splitlines()[-1])
print('DSRL_CONVT_CE=$v', d['value'], d['ms_per_step'], d['config']['losses_last_step'])
EOF
done
