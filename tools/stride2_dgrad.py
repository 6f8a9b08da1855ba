#!/usr/bin/env python3
"""The four stride-2 convs of ResNet-101 OS16 (layer2.0 / layer3.0: 3x3 conv2 and the 1x1 downsample): forward vs dgrad time. The dgrad
gathers with a divisibility test per tap and row, so 3/4 of its multiply-adds act on zeros unless the rows of a tile share their parity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import run
for name, shp in (('layer2.0.conv2 3x3 s2 128->128 @64x128', (8, 128, 64, 128, 128, 3, 2, 1, 1)), ('layer3.0.conv2 3x3 s2 256->256 @32x64', (8, 256, 32, 64, 256, 3, 2, 1, 1)),
                  ('layer2.0.downsample 1x1 s2 256->512 @64x128', (8, 256, 64, 128, 512, 1, 2, 0, 1)), ('layer3.0.downsample 1x1 s2 512->1024 @32x64', (8, 512, 32, 64, 1024, 1, 2, 0, 1)),
                  ('same 3x3 at stride 1 on the output size: 128->128 @32x64', (8, 128, 32, 64, 128, 3, 1, 1, 1)), ('256->256 @16x32', (8, 256, 16, 32, 256, 3, 1, 1, 1))):
    res = []
    for what in ('fwd', 'dgrad'):
        ms, tf = run(*shp, what)
        res.append(f'{what} {ms * 1e3:.1f} us ({tf:.0f} TF)')
    print(f'{name:52s} ' + '  '.join(res), flush=True)
