#!/bin/bash
mkdir -p gpurun_out/r04k
timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s > gpurun_out/r04k/fuzz_tests.log 2>&1
echo "tests rc=$?"; grep -v "^$" gpurun_out/r04k/fuzz_tests.log | cut -c1-700 | tail -40
