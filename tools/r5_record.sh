#!/bin/bash
# round 5 record: whole GPU suite, smoke, default bench line, kernel stats + PMC traffic, SQ counters of the step
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5f_tests.txt 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r5f_tests.txt; tail -2 gpurun_out/r5f_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5f_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r5f_smoke.txt
bash tools/round_record.sh r5f
bash tools/pmc_step.sh r5f > gpurun_out/r5f_pmc_step.txt 2>&1; tail -3 gpurun_out/r5f_pmc_step.txt
ls gpurun_out | grep r5f
