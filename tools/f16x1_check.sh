#!/bin/bash
# f16x1 (one fp16 MFMA per product): per-conv error vs fp64 on the precision test shapes, whole-step throughput at 256x512 and at config 5's size
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_hip_parity.py -x -q -k "test_conv_precision_modes and f16x1" -s 2>&1 | grep -E "f16x1|passed|failed|Error" | head -12
for m in f16x3 f16x1; do
  echo "== 256x512 $m"; DSRL_CONV_PRECISION=$([ $m = f16x3 ] && echo 4 || echo 5) python bench.py --steps 30 --warmup 8 --no-prof --no-cpu-baseline --no-config5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('losses_last_step'))"
  echo "== 512x1024 $m"; DSRL_CONV_PRECISION=$([ $m = f16x3 ] && echo 4 || echo 5) python bench.py --height 512 --width 1024 --steps 8 --warmup 4 --no-prof --no-cpu-baseline --no-config5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('losses_last_step'))"
done
