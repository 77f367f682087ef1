#!/bin/bash
mkdir -p gpurun_out/r04r
timeout 1500 python -m pytest tests/test_rebalance.py tests/test_bench_harness.py -m gpu -x -q -k "product_path or strong_scaling or rccl_branch" > gpurun_out/r04r/tests.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r04r/tests.log | cut -c1-600
bash tools/r04_leanmerge.sh
