#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -k "precision_modes or f16x3 or conv_golden or wgrad_group" > gpurun_out/r3e_tests_modes.txt 2>&1; echo "modes rc=$?"; tail -3 gpurun_out/r3e_tests_modes.txt
bash tools/r3_quick.sh r3e 4 || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3e_tests.txt 2>&1; echo "suite rc=$?"
tail -5 gpurun_out/r3e_tests.txt
