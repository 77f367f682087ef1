#!/bin/bash
# the whole GPU suite + smoke, then the Zipf counters
mkdir -p gpurun_out/r04s
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04s/tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r04s/tests.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/zipf_pmc.sh 1000000 > gpurun_out/r04s/zipf_pmc.txt 2>&1; tail -12 gpurun_out/r04s/zipf_pmc.txt
