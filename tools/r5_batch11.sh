#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5_b11_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b11_tests.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for r in 1 2 3; do
echo "A default (producers write into the concat buffers)  $(timeout -k 10 200 $B 2>>gpurun_out/r5_b11.err | val)"
echo "B DSRL_CAT_INPLACE=0                                  $(DSRL_CAT_INPLACE=0 timeout -k 10 200 $B 2>>gpurun_out/r5_b11.err | val)"
done
} > gpurun_out/r5_b11.txt 2>&1
