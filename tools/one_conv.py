#!/usr/bin/env python3
"""Runs one conv shape/pass repeatedly (for rocprofv3 --pmc). usage: one_conv.py <shape> <fwd|dgrad|wgrad> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sweep_conv import SHAPES, run
name, what = sys.argv[1], sys.argv[2]
ms, tf = run(*SHAPES[name], what)
print(name, what, f'{ms*1e3:.1f} us {tf:.1f} TF')
