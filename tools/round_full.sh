#!/bin/bash
# full GPU test suite, then the round record.  usage: tools/round_full.sh [tag, default r4]
tag=${1:-r4}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.txt 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/${tag}_tests.txt
[ $rc -ne 0 ] && exit 1
bash tools/round_record.sh ${tag}
