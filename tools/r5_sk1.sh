#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cooperative" > gpurun_out/sk1_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/sk1_tests.txt
for v in 1 0 1 0; do
  DSRL_SK_COOP1=$v timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-prof --no-cpu-baseline --no-config5 > gpurun_out/sk1_bench_$v.json 2> gpurun_out/sk1_bench_$v.err || exit 1
  python - <<EOF
import json
d=json.loads(open('gpurun_out/sk1_bench_$v.json').read().strip().splitlines()[-1])
print('DSRL_SK_COOP1=$v', d['value'], d['ms_per_step'], d['config']['losses_last_step'])
EOF
done
