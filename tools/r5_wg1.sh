#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "aspp or ASPP or full_model or head or slot or pool" > gpurun_out/wg1_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/wg1_tests.txt
bash tools/kstats_run.sh wg3 || exit 1
grep -i "ATen\|total kernel\|CUDAFunctor_add" gpurun_out/wg3_kstats.txt
python - <<'EOF'
import csv
rows=list(csv.DictReader(open('gpurun_out/wg3_stats/p_kernel_stats.csv')))
for r in rows:
    if 'CUDAFunctor_add' in r['Name'] or 'gap_bwd' in r['Name']: print(int(r['Calls'])/14, float(r['AverageNs'])/1e3, r['Name'][:90])
EOF
