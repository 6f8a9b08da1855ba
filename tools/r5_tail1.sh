#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convT" > gpurun_out/t1_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t1_tests.txt
timeout -k 10 200 python tools/convt_bench.py > gpurun_out/t1_convt.txt 2>&1; echo "bench rc=$?"; cat gpurun_out/t1_convt.txt
