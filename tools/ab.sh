#!/bin/bash
# A/B of two builds of the library on ONE box: ab/libdsrl_hip_prev.so (A) against the tree's build (B), alternating, default bench step.
# Extra arguments are environment settings applied to both (e.g. DSRL_WGRAD3_PX=2048).
mkdir -p gpurun_out
L=dualsuperreslearningforsemseg_amd/libdsrl_hip.so
cp $L /tmp/new.so
for r in 1 2; do
  cp ab/libdsrl_hip_prev.so $L
  echo "A $(env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
  cp /tmp/new.so $L
  echo "B $(env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
done
