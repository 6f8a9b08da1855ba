#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wgrad or conv_golden or full_model or head" > gpurun_out/wg1_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/wg1_tests.txt
for v in 1 0 1 0; do
  DSRL_WGRAD_IDENT=$v timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-prof --no-cpu-baseline --no-config5 > gpurun_out/wg1_bench_$v.json 2> gpurun_out/wg1_bench_$v.err || exit 1
  python - <<EOF
import json
d=json.loads(open('gpurun_out/wg1_bench_$v.json').read().strip().splitlines()[-1])
print('DSRL_WGRAD_IDENT=$v', d['value'], d['ms_per_step'], d['config']['losses_last_step'])
EOF
done
