#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
DSRL_CONV_PRECISION=4 timeout -k 10 300 python tools/count_calls.py > gpurun_out/r3b_calls_f16.txt 2>&1; echo "calls rc=$?"; head -30 gpurun_out/r3b_calls_f16.txt
for m in 2 4 2 4; do
  DSRL_CONV_PRECISION=$m timeout -k 10 300 python bench.py --no-prof --no-cpu-baseline --steps 40 --warmup 12 > gpurun_out/r3b_bench_m$m.txt 2>&1; echo "bench m$m rc=$?"
  python - <<PY
import json
for l in open('gpurun_out/r3b_bench_m$m.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('mode $m', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['losses_last_step'])
PY
done
DSRL_CONV_PRECISION=4 timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3b_tests_f16.txt 2>&1; echo "f16 suite rc=$?"
tail -5 gpurun_out/r3b_tests_f16.txt
