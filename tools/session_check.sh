#!/bin/bash
# suite + smoke + the bench as the driver runs it (stdout = the short line only):  gpurun -- bash tools/session_check.sh [out-name]
O=gpurun_out/${1:-check}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/bench_detail.json > $O/bench.out 2> $O/bench.err
echo "bench rc=$? stdout bytes $(wc -c < $O/bench.out) lines $(wc -l < $O/bench.out) stderr bytes $(wc -c < $O/bench.err)"
tail -c 3000 $O/bench.out
