#!/bin/bash
mkdir -p gpurun_out
DSRL_TAP_FAST=1 timeout -k 10 900 python -m pytest tests/test_hip_parity.py -q -k "conv_golden or precision_modes or f16x3 or real_shapes or strided_dgrad or channel_slices or epilogue or head_small" > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; grep "FAILED\|passed\|failed" gpurun_out/r3q_t.txt | head -20
bash tools/ab_env.sh DSRL_TAP_FAST=1 3
