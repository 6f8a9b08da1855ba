#!/bin/bash
# N > 1 path on the final build: real RCCL kernels on a 1-rank group claiming two ranks, then two gloo ranks sharing the one GPU through bench.py
mkdir -p gpurun_out
REH_STEPS=12 timeout -k 10 500 python tools/rccl_rehearsal.py > gpurun_out/sc_rccl.txt 2>&1; echo "rccl rehearsal rc=$?"; tail -4 gpurun_out/sc_rccl.txt
DSRL_ALL_RANKS_ON_GPU0=1 DSRL_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-config5 --no-prof > gpurun_out/sc_rehearsal.json 2> gpurun_out/sc_rehearsal.err; echo "gloo 2-rank bench rc=$?"
tail -c 900 gpurun_out/sc_rehearsal.json
