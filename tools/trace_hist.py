#!/usr/bin/env python3
"""Per (kernel, grid size) table of a rocprofv3 kernel-trace CSV: launches per step, mean duration, share - where the non-conv time of the step goes by
tensor size.  usage: trace_hist.py <p_kernel_trace.csv> <steps in the trace> [name regex] [top]"""
import csv, re, sys, collections
steps = float(sys.argv[2]); pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else '.'); top = int(sys.argv[4]) if len(sys.argv) > 4 else 60
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    name = r['Kernel_Name']
    if not pat.search(name): continue
    short = re.sub(r'\(.*', '', re.sub(r'^void |dsrl::', '', name))[:70]
    key = (short, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']))
    a = agg[key]; a[0] += 1; a[1] += d
print(f'all kernels: {tot / 1e3 / steps:.2f} ms per step; matched: {sum(v[1] for v in agg.values()) / 1e3 / steps:.2f} ms')
for (name, gx, gy), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{us / steps:8.1f} us/step  x{n / steps:6.1f}  {us / n:7.1f} us  grid {gx}x{gy}  {name}')
