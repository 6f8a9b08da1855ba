#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "presplit or transposes or filter or optimiser or magnitudes or train_steps or planes" > gpurun_out/ws_tests.txt 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/ws_tests.txt
bash tools/kstats_run.sh ws1 || exit 1
grep -i "weight_split\|total kernel\|SGD" gpurun_out/ws1_kstats.txt
python - <<'EOF'
import csv
for r in csv.DictReader(open('gpurun_out/ws1_stats/p_kernel_stats.csv')):
    if 'weight_split' in r['Name'] or 'filter_planes' in r['Name']: print(int(r['Calls'])/14, round(float(r['AverageNs'])/1e3,1), r['Name'][:70])
EOF
