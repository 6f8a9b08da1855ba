#!/bin/bash
# round 5, batch 2: new / changed GPU tests, the N > 1 schedule against real RCCL kernels (1-rank group) and as a 2-rank gloo rehearsal on one GPU
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "cooperative or frozen_bn or convT or planes_kernel" > gpurun_out/r5_b2_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r5_b2_tests.txt
timeout -k 10 600 python -m pytest tests/test_rccl_gpu.py -x -q -m gpu > gpurun_out/r5_b2_rccl.txt 2>&1; echo "rccl rc=$?" >> gpurun_out/r5_b2_rccl.txt
DSRL_ALL_RANKS_ON_GPU0=1 DSRL_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 6 --warmup 4 --no-cpu-baseline --no-config5 --no-prof > gpurun_out/r5_b2_rehearsal.json 2> gpurun_out/r5_b2_rehearsal.err; echo "rehearsal rc=$?" >> gpurun_out/r5_b2_rehearsal.err
tail -3 gpurun_out/r5_b2_tests.txt gpurun_out/r5_b2_rccl.txt; tail -2 gpurun_out/r5_b2_rehearsal.err
