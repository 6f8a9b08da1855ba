#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r3t -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 4 --no-prof --no-cpu-baseline --no-config5 > $R/gpurun_out/r3t.log 2>&1 || exit 1
python3 $R/tools/trace_gaps.py $R/gpurun_out/r3t/p_kernel_trace.csv > $R/gpurun_out/r3t_gaps.txt
python3 $R/tools/trace_hist.py $R/gpurun_out/r3t/p_kernel_trace.csv 14 'conv_igemm|conv_wgrad' 40 > $R/gpurun_out/r3t_hist_conv.txt
python3 $R/tools/trace_hist.py $R/gpurun_out/r3t/p_kernel_trace.csv 14 'bn_|amax|copy|at::|fill|bilinear|shuffle|pool|pointwise|dropout|colsum|sgd|weight_|ce_|mse_|fa_|convt|splitk|wgrad_reduce|zero_fill' 40 > $R/gpurun_out/r3t_hist.txt
rm -rf $R/gpurun_out/r3t
cat $R/gpurun_out/r3t_gaps.txt
