#!/bin/bash
# round 5, batch 5: fast BatchNorm-sum epilogue (tests + in step, with and without bn3's sums from the next block's accumulating dgrad), per-tap weight-gradient tiles
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_rccl_gpu.py -x -q -m gpu -k "planes_kernel or full_model or head_train or optimiser_pass or bottleneck or bn_bwd or dgrad" > gpurun_out/r5_b5_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b5_tests.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for r in 1 2; do
echo "A default                       $(timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
echo "B DSRL_BNSTATS_FAST=0           $(DSRL_BNSTATS_FAST=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
echo "C bn3 shared sums + fast        $(DSRL_BN_BWD_STATS_SHARED=1 timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
echo "D wgrad big 128x128             $(DSRL_WGRAD_BIG_CFG=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
echo "E wgrad big 256x64              $(DSRL_WGRAD_BIG_CFG=1 timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
echo "F wgrad big 64x128              $(DSRL_WGRAD_BIG_CFG=5 timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
echo "G shared + fast + wgrad 128x128 $(DSRL_BN_BWD_STATS_SHARED=1 DSRL_WGRAD_BIG_CFG=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b5.err | val)"
done
} > gpurun_out/r5_b5.txt 2>&1
