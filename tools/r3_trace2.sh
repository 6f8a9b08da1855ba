#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DSRL_WGRAD_PHASE_OVERLAP=1
rocprofv3 --kernel-trace -d $R/gpurun_out/r3t -o p --output-format csv -- python3 $R/bench.py --steps 6 --warmup 4 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/r3t.log 2>&1 || exit 1
python3 - <<'PY' > $R/gpurun_out/r3t_overlap.txt
import csv, os
rows = list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3t/p_kernel_trace.csv')))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')) for r in rows))
wg = [e for e in ev if 'conv_wgrad_group_kernel' in e[2]][-4:]
for s, e, n, q, st in wg:
    inside = [x for x in ev if x[0] < e and x[1] > s and x[2] != n]
    print(f'{n} queue {q} stream {st} {(e - s) / 1e3:.0f} us: {len(inside)} kernels overlap it, covering {sum(min(x[1], e) - max(x[0], s) for x in inside) / 1e3:.0f} us; queues {sorted(set(x[3] for x in inside))}')
    for x in inside[:5]: print('    ', x[2], (x[1] - x[0]) / 1e3)
PY
rm -rf $R/gpurun_out/r3t
cat $R/gpurun_out/r3t_overlap.txt
