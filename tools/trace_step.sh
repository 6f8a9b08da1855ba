#!/bin/bash
# Kernel trace of a few graph-replayed steps; prints the kernel sequence of the last step with durations (us), sorted as executed.
# usage (inside gpurun): bash tools/trace_step.sh [regex]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
GRAPH_ONLY=1 STEPS=4 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_step -o p --output-format csv -- python3 $R/tools/graph_probe.py > $R/gpurun_out/trace_step.log 2>&1 || exit 1
python3 - "$1" <<EOT
import csv, os, re, sys
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/trace_step/p_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = from the last rng_advance kernel on
idx = [i for i, r in enumerate(rows) if 'rng_advance' in r['Kernel_Name']]
last = rows[idx[-1]:]
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] else None
t0 = int(last[0]['Start_Timestamp'])
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/trace_step_last.txt', 'w')
for i, r in enumerate(last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    line = f"{i:4d} {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f} us  grid {r.get('Grid_Size', '?'):>9s}  {r['Kernel_Name'][:110]}"
    out.write(line + '\n')
    if pat is None or pat.search(r['Kernel_Name']): print(line)
print('kernels in the step:', len(last), ' span %.2f ms' % ((int(last[-1]['End_Timestamp']) - t0) / 1e6))
EOT
rm -rf $R/gpurun_out/trace_step
