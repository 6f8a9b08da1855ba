#!/usr/bin/env python3
"""A/B of env knobs on a few shapes: python tools/ab_conv.py KNOB=v1,v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import SHAPES, run
knob, vals = sys.argv[1].split('=')
vals = vals.split(',')
for name, shp in SHAPES.items():
    for what in ('fwd', 'dgrad', 'wgrad'):
        res = []
        for rep in range(2):
            for v in vals:
                os.environ[knob] = v
                ms, tf = run(*shp, what)
                res.append(f'{v}:{ms*1e3:.0f}us/{tf:.0f}TF')
        print(f'{name:10s} {what:6s} ' + '  '.join(res), flush=True)
