for px in 1536 2048 3072 4096 6144 8192 16384; do
  bash tools/kstats_run.sh px$px DSRL_WGRAD3_PX=$px || exit 1
  echo "px $px: $(grep 'conv_wgrad3_group_kernel<2, 1>' gpurun_out/px${px}_kstats.txt | cut -c1-32) | $(grep 'conv_wgrad3_group_kernel<2, 2>' gpurun_out/px${px}_kstats.txt | cut -c1-32) | $(grep 'slab reduces' gpurun_out/px${px}_kstats.txt | cut -c1-40) | $(head -1 gpurun_out/px${px}_kstats.txt)"
done
