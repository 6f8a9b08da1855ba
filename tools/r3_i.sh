#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -k "epilogue or statistics or strided_dgrad or presplit or head_256 or full_model" > gpurun_out/r3i_tests.txt 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3i_tests.txt
bash tools/r3_quick.sh r3i 4
