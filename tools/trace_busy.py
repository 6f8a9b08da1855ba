#!/usr/bin/env python3
"""GPU busy time (union of kernel intervals over all streams) vs wall time per training step, from a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
for s in range(max(1, len(idx) - 4), len(idx)):
    step = rows[idx[s - 1] + 1: idx[s] + 1]
    iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    wall = int(rows[idx[s]]['End_Timestamp']) - int(rows[idx[s - 1]]['End_Timestamp'])
    ksum = sum(b - a for a, b in iv)
    # idle before the first conv of the step and inside forward (until the CE loss kernel)
    ce = next(i for i, r in enumerate(step) if 'ce_fwd' in r['Kernel_Name'])
    fwd_wall = int(step[ce]['End_Timestamp']) - int(rows[idx[s - 1]]['End_Timestamp'])
    fwd_busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step[:ce + 1])
    print(f'step {s}: wall {wall/1e6:.2f} ms  busy(union) {busy/1e6:.2f}  idle {(wall-busy)/1e6:.2f}  kernel-sum {ksum/1e6:.2f}  kernels {len(step)} | '
          f'forward: wall {fwd_wall/1e6:.2f} busy {fwd_busy/1e6:.2f}')
