#!/usr/bin/env python3
"""Idle time between consecutive kernels of the replayed step in a rocprofv3 kernel trace: start(i+1) - end(i), per step and as a histogram.
usage: trace_gaps.py <p_kernel_trace.csv>"""
import csv, sys, collections
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))))
# steps: split at the SGD kernel
steps, cur = [], []
for r in rows:
    cur.append(r)
    if 'sgd' in r[2]:
        steps.append(cur); cur = []
steps = steps[4:]          # eager warm-up / capture iterations first
tot_gap = tot_busy = tot_wall = 0.0
hist = collections.Counter()
big = collections.Counter()
for st in steps:
    wall = (st[-1][1] - st[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in st) / 1e3
    gaps = [(st[i + 1][0] - st[i][1]) / 1e3 for i in range(len(st) - 1)]
    tot_gap += sum(g for g in gaps if g > 0); tot_busy += busy; tot_wall += wall
    for i, g in enumerate(gaps):
        hist[min(int(g), 10)] += 1
        if g > 3: big[(st[i][2][:40], st[i + 1][2][:40])] += g
n = len(steps)
print(f'{n} replayed steps: wall {tot_wall / n / 1e3:.2f} ms, kernels busy {tot_busy / n / 1e3:.2f} ms, idle between kernels {tot_gap / n / 1e3:.2f} ms ({len(steps[0])} kernels per step)')
print('gap histogram (us, floor; 10 = 10 or more):', sorted(hist.items()))
for k, v in big.most_common(8): print(f'  {v / n:7.1f} us/step  {k[0]} -> {k[1]}')
