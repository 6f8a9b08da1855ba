#!/bin/bash
# round 5, batch 3: in-step A/B of the cooperative split-K planner rule and of the per-tap weight-gradient tile knobs
mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for r in 1 2; do
echo "A SK_AUTO=0 (round 4 plan)   $(DSRL_SK_AUTO=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b3.err | val)"
echo "B SK_AUTO=1 (layer4 coop)    $(DSRL_SK_AUTO=1 timeout -k 10 200 $B 2>>gpurun_out/r5_b3.err | val)"
echo "C SK_AUTO=2 (+layer3 coop)   $(DSRL_SK_AUTO=2 timeout -k 10 200 $B 2>>gpurun_out/r5_b3.err | val)"
done
echo "D wgrad big cfg 128x128      $(DSRL_WGRAD_BIG_CFG=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b3.err | val)"
echo "E wgrad group px 2048        $(DSRL_WGRAD_GROUP_PX=2048 timeout -k 10 200 $B 2>>gpurun_out/r5_b3.err | val)"
echo "F wgrad group px 8192        $(DSRL_WGRAD_GROUP_PX=8192 timeout -k 10 200 $B 2>>gpurun_out/r5_b3.err | val)"
} > gpurun_out/r5_b3.txt 2>&1
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cooperative or frozen_bn" > gpurun_out/r5_b3_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b3_tests.txt
