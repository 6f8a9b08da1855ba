#!/bin/bash
# round 5, batch 1: BatchNorm prologue depth (isolated + in step), bn3 backward sums from the next block's accumulating dgrad, cooperative split-K on layer4
mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
echo "== bn_fwd_bench U=4"; DSRL_BN_PROLOGUE_U=4 timeout -k 10 120 python tools/bn_fwd_bench.py 2>/dev/null
echo "== bn_fwd_bench U=16"; timeout -k 10 120 python tools/bn_fwd_bench.py 2>/dev/null
for r in 1 2; do
echo "A default            $(timeout -k 10 200 $B 2>>gpurun_out/r5_b1.err | val)"
echo "B prologue U=4       $(DSRL_BN_PROLOGUE_U=4 timeout -k 10 200 $B 2>>gpurun_out/r5_b1.err | val)"
echo "C bn3 shared sums    $(DSRL_BN_BWD_STATS_SHARED=1 timeout -k 10 200 $B 2>>gpurun_out/r5_b1.err | val)"
done
} > gpurun_out/r5_b1.txt 2>&1
