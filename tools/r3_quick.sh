#!/bin/bash
# quick A/B of the graph-replayed step: tools/r3_quick.sh <tag> <modes...>   (bench --no-prof, 40 steps each, then rocprof stats of the first mode)
tag=$1; shift
mkdir -p gpurun_out
for m in "$@" "$@"; do
  DSRL_CONV_PRECISION=$m timeout -k 10 300 python bench.py --no-prof --no-cpu-baseline --no-config5 --steps 40 --warmup 12 > gpurun_out/${tag}_bench_m$m.txt 2>&1 || { echo "bench m$m failed"; tail -5 gpurun_out/${tag}_bench_m$m.txt; exit 1; }
  python - <<PY
import json
for l in open('gpurun_out/${tag}_bench_m$m.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('mode $m', d['value'], 'img/s', d['ms_per_step'], 'ms', d['config']['losses_last_step'])
PY
done
bash tools/r3_prof.sh ${tag} $1 > /dev/null && head -20 gpurun_out/${tag}_kstats.txt
