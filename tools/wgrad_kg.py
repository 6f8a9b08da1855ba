#!/usr/bin/env python3
"""wgrad time per layer shape for the pixel-group counts 1 / 2 / 4 (DSRL_FORCE_WGRAD_KG) and the library's own pick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import SHAPES, run
SHAPES['l4_3x3'] = (8, 512, 16, 32, 512, 3, 1, 2, 2)
SHAPES['l4_1x1_up'] = (8, 512, 16, 32, 2048, 1, 1, 0, 1)
SHAPES['l4_1x1_dn'] = (8, 2048, 16, 32, 512, 1, 1, 0, 1)
SHAPES['aspp_1x1'] = (8, 2048, 16, 32, 256, 1, 1, 0, 1)
SHAPES['l2_1x1_up'] = (8, 128, 32, 64, 512, 1, 1, 0, 1)
SHAPES['l2_1x1_dn'] = (8, 512, 32, 64, 128, 1, 1, 0, 1)
SHAPES['l1_1x1_up'] = (8, 64, 64, 128, 256, 1, 1, 0, 1)
SHAPES['l1_1x1_dn'] = (8, 256, 64, 128, 64, 1, 1, 0, 1)
SHAPES['cat4'] = (8, 256, 64, 128, 256, 3, 1, 1, 1)
SHAPES['sisr'] = (8, 304, 64, 128, 192, 3, 1, 1, 1)
for name, shp in SHAPES.items():
    res = []
    for kg in (0, 1, 2, 4):
        if kg:
            os.environ['DSRL_FORCE_WGRAD_KG'] = str(kg)
        else:
            os.environ.pop('DSRL_FORCE_WGRAD_KG', None)
        ms, tf = run(*shp, 'wgrad')
        res.append(f"{'auto' if not kg else 'kg' + str(kg)}:{ms*1e3:.0f}")
    print(f'{name:10s} wgrad ' + '  '.join(res), flush=True)
