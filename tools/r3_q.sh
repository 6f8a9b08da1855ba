#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r3q_t.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3q_t.txt
[ $rc -ne 0 ] && exit 1
bash tools/ab.sh
