#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convT or logits_gradient or fused_losses or total_loss_backward" > gpurun_out/t3_tests.txt 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/t3_tests.txt
