#!/bin/bash
# end-of-round check: smoke(), the whole GPU suite, then the record
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3_smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r3_smoke.txt
bash tools/round_full.sh
