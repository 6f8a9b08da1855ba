#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "head or dropout or full_model or planes or train_step" > gpurun_out/r5_b10_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b10_tests.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "A default (decoder BN links)    $(timeout -k 10 200 $B 2>>gpurun_out/r5_b10.err | val)"
echo "B DSRL_BN_BWD_STATS=0           $(DSRL_BN_BWD_STATS=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b10.err | val)"
done
} > gpurun_out/r5_b10.txt 2>&1
bash tools/kstats_run.sh r5_dec
