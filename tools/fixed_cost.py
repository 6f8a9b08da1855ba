#!/usr/bin/env python3
"""Fixed cost of a conv launch: the layer3 output shape (M = 4096 pixels) with K shrunk to one 32-channel chunk, next to the real layers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import run
for name, shp in (('1x1 C=32 -> 256', (8, 32, 16, 32, 256, 1, 1, 0, 1)), ('1x1 C=32 -> 1024', (8, 32, 16, 32, 1024, 1, 1, 0, 1)),
                  ('1x1 C=256 -> 256', (8, 256, 16, 32, 256, 1, 1, 0, 1)), ('1x1 C=1024 -> 256', (8, 1024, 16, 32, 256, 1, 1, 0, 1)),
                  ('1x1 C=256 -> 1024', (8, 256, 16, 32, 1024, 1, 1, 0, 1)), ('3x3 C=32 -> 256', (8, 32, 16, 32, 256, 3, 1, 1, 1)),
                  ('3x3 C=256 -> 256', (8, 256, 16, 32, 256, 3, 1, 1, 1))):
    res = []
    for what in ('fwd', 'dgrad', 'wgrad'):
        ms, tf = run(*shp, what)
        res.append(f'{what} {ms*1e3:.1f}us')
    print(f'{name:20s} ' + '  '.join(res), flush=True)
