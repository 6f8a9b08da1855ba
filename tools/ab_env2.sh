#!/bin/bash
# several environment variants against the default on ONE box, alternating: tools/ab_env2.sh "VAR=1 VAR2=2" "VAR3=0" ...
b() { echo "$(env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-100)"; }
for r in 1 2; do
  echo "default: $(b A=1)"
  for v in "$@"; do echo "$v: $(b $v)"; done
done
