#!/bin/bash
# the one-group cooperative split-K plan (DSRL_SK_COOP1=1) through the whole GPU suite and the full bench command (all arithmetic modes + the 512x1024 size)
mkdir -p gpurun_out
export DSRL_SK_COOP1=1
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/c1_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/c1_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > gpurun_out/c1_bench.json 2> gpurun_out/c1_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c1_bench.json; tail -3 gpurun_out/c1_bench.err | cut -c1-200
