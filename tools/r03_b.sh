#!/bin/bash
# history-slot dispatch in the team kernels: parity of every team routing, then the Zipf shapes
mkdir -p gpurun_out/r03b
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -k "block_kernel or team_tiers or device_wide or large_and_giant or c5_share or fixed" > gpurun_out/r03b/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03b/tests.log
for w in zipf c5share; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03b/bench_$w.json 2> gpurun_out/r03b/bench_$w.err
  echo "$w rc=$?"
done
