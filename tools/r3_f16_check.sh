#!/bin/bash
# round 3: first GPU check of the f16x3 arithmetic (accuracy against fp64, timing against the other modes, parity tests in that mode)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -k "precision_modes or f16x3 or conv_golden" > gpurun_out/r3a_tests_modes.txt 2>&1; echo "modes rc=$?" | tee -a gpurun_out/r3a_tests_modes.txt
tail -5 gpurun_out/r3a_tests_modes.txt
timeout -k 10 400 python tools/prec_check.py > gpurun_out/r3a_prec_check.txt 2>&1; echo "prec rc=$?"
PREC_DY=tiny timeout -k 10 400 python tools/prec_check.py > gpurun_out/r3a_prec_check_tiny.txt 2>&1; echo "prec tiny rc=$?"
DSRL_CONV_PRECISION=4 timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3a_tests_f16.txt 2>&1; echo "f16 suite rc=$?"
tail -15 gpurun_out/r3a_tests_f16.txt
