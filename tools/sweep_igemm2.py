#!/usr/bin/env python3
"""Forward / dgrad time per layer shape over tile config x (split-K | K groups), current arithmetic mode; 'auto' = the library's own pick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import SHAPES, run
names = ['T128x128', 'T256x64', 'T256x32', 'T64x64', 'T128x64', 'T64x128', 'T128x32', 'T256x128', 'T256x256']
SHAPES['l4_3x3'] = (8, 512, 16, 32, 512, 3, 1, 2, 2)
SHAPES['l4_1x1_up'] = (8, 512, 16, 32, 2048, 1, 1, 0, 1)
SHAPES['l4_1x1_dn'] = (8, 2048, 16, 32, 512, 1, 1, 0, 1)
SHAPES['aspp_d6'] = (8, 2048, 16, 32, 256, 3, 1, 6, 6)
SHAPES['aspp_1x1'] = (8, 2048, 16, 32, 256, 1, 1, 0, 1)
SHAPES['l2_1x1_up'] = (8, 128, 32, 64, 512, 1, 1, 0, 1)
SHAPES['l2_1x1_dn'] = (8, 512, 32, 64, 128, 1, 1, 0, 1)
SHAPES['l1_1x1_up'] = (8, 64, 64, 128, 256, 1, 1, 0, 1)
SHAPES['l1_1x1_dn'] = (8, 256, 64, 128, 64, 1, 1, 0, 1)
SHAPES['shortcut'] = (8, 256, 64, 128, 48, 1, 1, 0, 1)
SHAPES['cat4'] = (8, 256, 64, 128, 256, 3, 1, 1, 1)
SHAPES['sisr'] = (8, 304, 64, 128, 192, 3, 1, 1, 1)
auto_only = '--auto' in sys.argv
which = [a for a in sys.argv[1:] if not a.startswith('--')] or list(SHAPES)
for name in which:
    for what in ('fwd', 'dgrad'):
        for k in ('DSRL_FORCE_CFG', 'DSRL_FORCE_SPLITS', 'DSRL_FORCE_KG'):
            os.environ.pop(k, None)
        ms, tf = run(*SHAPES[name], what)
        line = [f'auto:{ms*1e3:.0f}']
        if auto_only:
            print(f'{name:10s} {what:5s} ' + line[0], flush=True)
            continue
        for cfg in [int(c) for c in os.environ.get('SWEEP_CFGS', '0,1,3,4,5').split(',')]:
            os.environ['DSRL_FORCE_CFG'] = str(cfg)
            res = []
            for kg, sp in (((1, 1), (1, 2), (1, 4), (1, 8)) if cfg >= 7 else ((1, 1), (1, 2), (1, 4), (2, 1), (4, 1))):
                os.environ['DSRL_FORCE_KG'] = str(kg); os.environ['DSRL_FORCE_SPLITS'] = str(sp)
                ms, tf = run(*SHAPES[name], what)
                res.append(f'{ms*1e3:.0f}')
            line.append(f'{names[cfg]}[s1,s2,s4,g2,g4]=' + '/'.join(res))
        print(f'{name:10s} {what:5s} ' + '  '.join(line), flush=True)
