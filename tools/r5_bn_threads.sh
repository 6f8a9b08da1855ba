#!/bin/bash
# from-statistics BatchNorm kernels: block size A/B on one box (DSRL_BN_APPLY_THREADS; unset = chosen per launch), merge probe + default bench step, alternating
for t in 256 512 1024 0; do
  echo "== merge probe, $t threads (0: per launch)"; DSRL_BN_APPLY_THREADS=$t timeout -k 10 200 python tools/bn_prologue_probe.py 2>/dev/null
done
for r in 1 2 3; do
  for t in 256 512 1024 0; do
    echo "threads $t  $(DSRL_BN_APPLY_THREADS=$t timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
  done
done
