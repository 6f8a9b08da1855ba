#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -k "precision_modes or f16x3 or conv_golden or wgrad_group or bn_backward_statistics" > gpurun_out/r3g_tests_modes.txt 2>&1; echo "modes rc=$?"; tail -4 gpurun_out/r3g_tests_modes.txt
for v in 1 0 1 0; do
  DSRL_PRESPLIT=$v timeout -k 10 300 python bench.py --no-prof --no-cpu-baseline --steps 40 --warmup 12 > gpurun_out/r3g_bench_p$v.txt 2>&1 || { echo "bench failed"; tail -5 gpurun_out/r3g_bench_p$v.txt; exit 1; }
  python - <<PY
import json
for l in open('gpurun_out/r3g_bench_p$v.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('presplit $v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['losses_last_step'])
PY
done
bash tools/r3_prof.sh r3g 4 > /dev/null && head -8 gpurun_out/r3g_kstats.txt
