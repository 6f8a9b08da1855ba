#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_rccl_gpu.py -x -q > gpurun_out/r3j_rccl.txt 2>&1; echo "rccl rc=$?"; tail -15 gpurun_out/r3j_rccl.txt
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -k "fused_losses or train_or_resume" > gpurun_out/r3j_t2.txt 2>&1; echo "t2 rc=$?"; tail -3 gpurun_out/r3j_t2.txt
# two real ranks sharing the one GPU over gloo: self-launch, split capture on both ranks, collectives between / behind the graphs, JSON line
DSRL_ALL_RANKS_ON_GPU0=1 DSRL_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 6 --warmup 4 --batch 2 --no-prof --no-cpu-baseline > gpurun_out/r3j_gloo2.txt 2>&1; echo "gloo2 rc=$?"; tail -3 gpurun_out/r3j_gloo2.txt | cut -c1-1500
