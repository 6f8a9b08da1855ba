#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import SHAPES, run
SHAPES['l4_3x3'] = (8, 512, 16, 32, 512, 3, 1, 2, 2)
SHAPES['aspp_d6'] = (8, 2048, 16, 32, 256, 3, 1, 6, 6)
SHAPES['l4_1x1_up'] = (8, 512, 16, 32, 2048, 1, 1, 0, 1)
SHAPES['sisr'] = (8, 304, 64, 128, 192, 3, 1, 1, 1)
names = ['T128x128', 'T256x64', 'T256x32', 'T64x64', 'T128x64', 'T64x128', 'T128x32']
for name in ('l3_3x3', 'l3_1x1_up', 'l3_1x1_dn', 'l2_3x3', 'l1_3x3', 'l4_3x3', 'l4_1x1_up', 'aspp_d6', 'cat0', 'sisr'):
    for cfg in (0, 1, 3, 4, 5):
        os.environ['DSRL_FORCE_CFG'] = str(cfg)
        res = []
        for sp in (4, 8, 16, 32, 64, 128):
            os.environ['DSRL_FORCE_PSPLITS'] = str(sp)
            ms, tf = run(*SHAPES[name], 'wgrad')
            res.append(f'{sp}:{ms*1e3:.0f}us/{tf:.0f}TF')
        print(f'{name:10s} wgrad {names[cfg]:9s} ' + '  '.join(res), flush=True)
