#!/bin/bash
mkdir -p gpurun_out
K="conv_golden or precision_modes or f16x3 or strided_dgrad or real_shapes or channel_slices or epilogue or rowfold"
for c in 7 8; do
DSRL_FORCE_CFG=$c timeout -k 10 400 python -m pytest tests/test_hip_parity.py -x -q -k "$K" > gpurun_out/r3q_t$c.txt 2>&1; rc=$?; echo "forced cfg $c rc=$rc"; tail -3 gpurun_out/r3q_t$c.txt
[ $rc -ne 0 ] && exit 1
done
for sh in aspp_d6 l4_3x3; do
for v in "0 8" "0 16" "7 8" "7 16" "7 4"; do set -- $v
echo "$sh cfg $1 splits $2: $(DSRL_FORCE_CFG=$1 DSRL_FORCE_SPLITS=$2 python tools/sweep_igemm2.py $sh --auto 2>/dev/null | tr '\n' ' ')"
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > gpurun_out/r3q_bench.txt 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r3q_bench.txt | cut -c1-400
DSRL_BIG_TILES=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 --no-prof > gpurun_out/r3q_bench0.txt 2>&1; echo "bench0 rc=$?"; tail -1 gpurun_out/r3q_bench0.txt | cut -c1-200
