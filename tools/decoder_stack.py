#!/usr/bin/env python3
"""bench.py's decoder-stack roofline alone (north_star: >= 30 % of the MFMA peak on the decoder conv stack): tools/decoder_stack.py [VAR=value ...]"""
import json, os, sys
for kv in sys.argv[1:]:
    k, v = kv.split('=', 1)
    os.environ[k] = v
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualsuperreslearningforsemseg_amd import functional as HF
HF.set_conv_precision('f16x3')
d = bench.decoder_stack_roofline(torch, HF, 8, 256, 512)
print(' '.join(sys.argv[1:]) or 'default', {k: (d[k]['frac'] if 'frac' in d[k] else d[k]['frac_of_time_weighted_peak'], d[k]['ms']) for k in ('forward', 'dgrad', 'wgrad', 'wgrad_per_layer_launches', 'all_passes')})
print('   wgrad TF/s per layer', {n: l['wgrad_per_layer_tflops'] for n, l in d['layers'].items()})
