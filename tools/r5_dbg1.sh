#!/bin/bash
T="python -m pytest tests/test_rccl_gpu.py -x -q -m gpu -k reduction_paths"
for kv in "X=1" "DSRL_BN_BWD_STATS_SHARED=0" "DSRL_BNSTATS_FAST=0" "DSRL_WGRAD_BIG_CFG=-1" "DSRL_SGD_AMAX=0" "DSRL_SK_AUTO=0"; do
  echo "== $kv: $(env $kv timeout -k 10 300 $T 2>&1 | tail -1)"
done
