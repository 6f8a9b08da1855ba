#!/bin/bash
# round 5, batch 8: one-plane (f16x1) operands through the LDS-DMA kernel in the step: tests, then A/B at both sizes
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "f16x1 or planes or precision" > gpurun_out/r5_b8_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b8_tests.txt
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
C="python bench.py --height 512 --width 1024 --steps 10 --warmup 5 --no-cpu-baseline --no-config5 --no-prof"
{
for r in 1 2; do
echo "256x512  f16x1 planes auto  $(DSRL_CONV_PRECISION=5 timeout -k 10 300 $B 2>>gpurun_out/r5_b8.err | val)"
echo "256x512  f16x1 planes off   $(DSRL_CONV_PRECISION=5 DSRL_PLANES_MODE=off timeout -k 10 300 $B 2>>gpurun_out/r5_b8.err | val)"
echo "256x512  f16x3 default      $(timeout -k 10 300 $B 2>>gpurun_out/r5_b8.err | val)"
echo "512x1024 f16x1 planes auto  $(DSRL_CONV_PRECISION=5 timeout -k 10 400 $C 2>>gpurun_out/r5_b8.err | val)"
echo "512x1024 f16x1 planes off   $(DSRL_CONV_PRECISION=5 DSRL_PLANES_MODE=off timeout -k 10 400 $C 2>>gpurun_out/r5_b8.err | val)"
echo "512x1024 f16x3 default      $(timeout -k 10 400 $C 2>>gpurun_out/r5_b8.err | val)"
done
} > gpurun_out/r5_b8.txt 2>&1
