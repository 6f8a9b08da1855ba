#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "conv_golden or precision_modes or f16x3 or real_shapes or head_small" > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3q_t.txt
[ $rc -ne 0 ] && exit 1
bash tools/ab.sh
