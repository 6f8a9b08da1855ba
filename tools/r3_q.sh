#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "conv_golden or precision_modes or f16x3 or nan_at_first or real_shapes or strided_dgrad or channel_slices or epilogue or head_small or head_train or full_model_vs_oracle" > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3q_t.txt
[ $rc -ne 0 ] && exit 1
bash tools/ab.sh
