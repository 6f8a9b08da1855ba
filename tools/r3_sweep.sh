#!/bin/bash
# f16x3 re-tuning: forward / dgrad tile x split plan per layer shape, and the grouped weight-gradient tile through the whole step
mkdir -p gpurun_out
timeout -k 10 900 python tools/sweep_igemm2.py l3_3x3 l3_1x1_up l3_1x1_dn l4_3x3 l4_1x1_up l4_1x1_dn aspp_1x1 l2_3x3 l2_1x1_up l2_1x1_dn l1_3x3 l1_1x1_up l1_1x1_dn cat0 cat4 sisr > gpurun_out/r3h_sweep_igemm.txt 2>&1
tail -40 gpurun_out/r3h_sweep_igemm.txt
for env in "X=0" "DSRL_WGRAD_BIG_CFG=0" "DSRL_WGRAD_BIG_CFG=5" "DSRL_WGRAD_BIG_CFG=1" "DSRL_WGRAD_GROUP_PX=2048" "DSRL_WGRAD_GROUP_PX=8192" "DSRL_WGRAD_BIG_CFG=0 DSRL_WGRAD_GROUP_PX=8192"; do
  env $env timeout -k 10 300 python bench.py --no-prof --no-cpu-baseline --steps 40 --warmup 12 > gpurun_out/r3h_b.txt 2>&1
  python - <<PY
import json
for l in open('gpurun_out/r3h_b.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('$env', d['value'], 'img/s', d['ms_per_step'], 'ms')
PY
done
