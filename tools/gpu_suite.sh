#!/bin/bash
# the whole GPU suite + smoke
mkdir -p gpurun_out/suite
cd /root/repo
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/suite/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/suite/tests.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
