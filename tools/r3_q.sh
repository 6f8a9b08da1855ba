#!/bin/bash
SWEEP_CFGS=0,1,7,8 timeout -k 10 600 python tools/sweep_igemm2.py cat0 cat4 sisr aspp_d6 l4_3x3 l4_1x1_up > gpurun_out/r3q_sweep2.txt 2>&1
grep dgrad gpurun_out/r3q_sweep2.txt | cut -c1-250
