#!/bin/bash
bash tools/kstats_run.sh t6a DSRL_CONVT_CE=1 || exit 1
grep -i "convt\|ce_fused\|loss pass\|ConvTranspose\|total kernel\|count_valid\|ce_final\|pointwise" gpurun_out/t6a_kstats.txt
bash tools/kstats_run.sh t6b DSRL_CONVT_CE=0 || exit 1
grep -i "convt\|ce_fused\|loss pass\|ConvTranspose\|total kernel\|count_valid\|ce_final\|pointwise" gpurun_out/t6b_kstats.txt
