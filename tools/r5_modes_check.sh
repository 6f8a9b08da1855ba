#!/bin/bash
# the GPU suite under the non-default plane modes and with the late-round knobs off, on the final build
mkdir -p gpurun_out
for cfg in "${@:-DSRL_PLANES_MODE=all}" ; do
  tag=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/modes_$tag.txt 2>&1; echo "$cfg rc=$? $(tail -1 gpurun_out/modes_$tag.txt)"
done
