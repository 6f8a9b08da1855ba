#!/bin/bash
# A/B of an environment switch on ONE box, alternating: tools/ab_env.sh VAR=value [rounds]
kv=$1; n=${2:-3}
for r in $(seq $n); do
  echo "A          $(timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
  echo "B $kv $(env $kv timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
done
