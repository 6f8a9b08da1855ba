#!/bin/bash
L=dualsuperreslearningforsemseg_amd/libdsrl_hip.so
cp $L /tmp/new.so
cp ab/libdsrl_hip_prev.so $L
echo "== prev library"; timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -s -m gpu -k "total_loss_backward_with_fa" 2>&1 | grep -E "^\{|passed|failed|Assertion"
cp /tmp/new.so $L
echo "== new library"; timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -s -m gpu -k "total_loss_backward_with_fa" 2>&1 | grep -E "^\{|passed|failed|Assertion"
