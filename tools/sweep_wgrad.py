#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import SHAPES, run
names = ['T128x128', 'T256x64', 'T256x32', 'T64x64', 'T128x64', 'T64x128', 'T128x32']
for name in ('l3_3x3', 'l3_1x1_up', 'l3_1x1_dn', 'l2_3x3', 'l1_3x3', 'cat0'):
    for cfg in (0, 3, 4, 5):
        os.environ['DSRL_FORCE_CFG'] = str(cfg)
        res = []
        for sp in (1, 2, 4, 8, 16, 32, 64):
            os.environ['DSRL_FORCE_PSPLITS'] = str(sp)
            ms, tf = run(*SHAPES[name], 'wgrad')
            res.append(f'{sp}:{ms*1e3:.0f}us/{tf:.0f}TF')
        print(f'{name:10s} wgrad {names[cfg]:9s} ' + '  '.join(res), flush=True)
