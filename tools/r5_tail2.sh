#!/bin/bash
mkdir -p gpurun_out
for cfg in "DSRL_CONVT_DMA=1" "DSRL_CONVT_ABL=2" "DSRL_CONVT_ABL=3" "DSRL_CONVT_ABL=4" "DSRL_CONVT_ABL=5"; do
  echo "== $cfg"; timeout -k 10 120 python tools/convt_kernel_bench.py $cfg 2>&1 | grep -v amdgpu.ids || exit 1
done
