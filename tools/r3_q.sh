#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3q_t.txt
[ $rc -ne 0 ] && exit 1
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130; done
bash tools/trace_step.sh zzzz | tail -1
