#!/bin/bash
# round 5, batch 4: whole GPU suite on the tree so far; A/B of the optimiser-pass magnitudes and of the per-tap weight-gradient tile
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5_b4_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b4_tests.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "A default (SGD leaves amax)   $(timeout -k 10 200 $B 2>>gpurun_out/r5_b4.err | val)"
echo "B DSRL_SGD_AMAX=0             $(DSRL_SGD_AMAX=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b4.err | val)"
echo "C wgrad big cfg 128x128       $(DSRL_WGRAD_BIG_CFG=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b4.err | val)"
done
} > gpurun_out/r5_b4.txt 2>&1
