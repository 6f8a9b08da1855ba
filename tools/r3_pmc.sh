#!/bin/bash
# SQ counters of single conv launches in the current default arithmetic
for sh in cat0 l3_3x3 l3_1x1_up; do for w in fwd dgrad wgrad; do bash tools/pmc_conv.sh r3_${sh}_$w $sh $w 2>&1 | tail -1; done; done
