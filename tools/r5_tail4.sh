#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convT or logits_gradient or loss or ce_ or cross" > gpurun_out/t3_tests.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t3_tests.txt
for v in 1 0 1 0; do
  DSRL_CONVT_CE=$v timeout -k 10 300 python bench.py --steps 60 --warmup 15 --no-prof --no-cpu-baseline --no-config5 > gpurun_out/t4_bench_$v.json 2> gpurun_out/t4_bench_$v.err || exit 1
  python - <<EOF
import json
d=json.loads(open('gpurun_out/t4_bench_$v.json').read().strip().splitlines()[-1])
print('DSRL_CONVT_CE=$v', d['value'], d['ms_per_step'], d['config']['losses_last_step'])
EOF
done
