#!/bin/bash
# the whole GPU suite + smoke
mkdir -p gpurun_out/r03s
cd /root/repo
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r03s/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03s/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
