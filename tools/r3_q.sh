#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "conv_golden or precision_modes or f16x3 or strided_dgrad or real_shapes or channel_slices or epilogue or bn_backward_statistics or grad_slots or head_small or full_model_vs_oracle or head_train" > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3q_t.txt
[ $rc -ne 0 ] && exit 1
python tools/epi_cost.py 2>&1 | grep -v amdgpu
bash tools/ab.sh
