#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convT" > gpurun_out/t3_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t3_tests.txt
for wv in 8 12; do
  echo "== waves $wv"; timeout -k 10 120 python tools/convt_kernel_bench.py DSRL_CONVT_CE_WAVES=$wv 2>&1 | grep -v amdgpu.ids | sed 's/.*inside//' || exit 1
done
