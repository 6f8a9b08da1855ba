#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3 4; do
DSRL_ALL_RANKS_ON_GPU0=1 DSRL_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 4 --warmup 4 --batch 2 --no-prof --no-cpu-baseline > gpurun_out/r3l_gloo$i.txt 2>&1; echo "gloo2 rc=$? $(grep -c 'AccumulateGrad node' gpurun_out/r3l_gloo$i.txt) warnings $(grep -o '"losses_last_step": \[[^]]*\]' gpurun_out/r3l_gloo$i.txt)"
done
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r3l_tests.txt 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r3l_tests.txt
bash tools/r3_quick.sh r3l 4 | head -4
