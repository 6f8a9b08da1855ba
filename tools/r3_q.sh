#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "batchnorm or bn_ or epilogue or strided_dgrad or full_model_vs_oracle or head_small" > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3q_t.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-prof > gpurun_out/r3q_bench.txt 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r3q_bench.txt | cut -c1-200
