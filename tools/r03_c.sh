#!/bin/bash
# tall kernel: parity on every fixture, C3 at full size, the ML-20M benches
mkdir -p gpurun_out/r03c
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -x -k "tall or default_routing or c3_at_full or reproducible" > gpurun_out/r03c/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03c/tests.log
for w in ml20m_user ml20m_movie; do
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03c/bench_$w.json 2> gpurun_out/r03c/bench_$w.err
  echo "$w rc=$?"
done
