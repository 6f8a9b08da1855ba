#!/bin/bash
bash tools/kstats_run.sh wg2a DSRL_WGRAD_IDENT=1 || exit 1
grep -i "wgrad\|total kernel" gpurun_out/wg2a_kstats.txt
bash tools/kstats_run.sh wg2b DSRL_WGRAD_IDENT=0 || exit 1
grep -i "wgrad\|total kernel" gpurun_out/wg2b_kstats.txt
