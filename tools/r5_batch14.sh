#!/bin/bash
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-config5 --no-prof"
val() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for r in 1 2 3; do
echo "A default (SK_AUTO=1)  $(timeout -k 10 200 $B 2>/dev/null | val)"
echo "B SK_AUTO=0            $(DSRL_SK_AUTO=0 timeout -k 10 200 $B 2>/dev/null | val)"
done
