#!/bin/bash
# Same-box comparison of the shadow weight-gradient schedule (DSRL_WGRAD_SHADOW=<resident blocks>) with the plain one.
# usage: tools/shadow_ab.sh "VAR=v VAR2=v" "VAR=v" ...   (one bench run per argument, environment as given)
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 240 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-prof --no-config5 2>gpurun_out/shadow_last.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'])
" || exit 1
done
