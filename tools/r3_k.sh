#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8; do
DSRL_DEBUG_SPLIT=1 DSRL_ALL_RANKS_ON_GPU0=1 DSRL_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 4 --warmup 4 --batch 2 --no-prof --no-cpu-baseline > gpurun_out/r3k_dbg$i.txt 2>&1; echo "f16 rc=$?"
grep "NaN params" gpurun_out/r3k_dbg$i.txt | head -1 | cut -c1-300
done
