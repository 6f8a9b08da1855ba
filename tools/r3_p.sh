#!/bin/bash
mkdir -p gpurun_out
for env in "X=0" "DSRL_BN_BWD_STATS_SHARED=1" "DSRL_BN_FUSED_BIG=0" "X=1" "DSRL_BN_BWD_STATS_SHARED=1"; do
  env $env timeout -k 10 300 python bench.py --no-prof --no-cpu-baseline --no-config5 --steps 40 --warmup 12 > gpurun_out/r3p_b.txt 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r3p_b.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('$env', d['value'], 'img/s', d['ms_per_step'], 'ms')
PY
done
