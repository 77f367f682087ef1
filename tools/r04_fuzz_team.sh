#!/bin/bash
# randomised parity sweep on the library with the tall team class and the spread launch plan (seeds disjoint from the earlier sweeps)
mkdir -p gpurun_out/r04g
cd /root/repo
export PYTHONPATH=.:tests
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "tallest or tall_team" 2>&1 | tail -3
( python tools/fuzz_parity.py ${1:-2000} ${2:-1300000} > gpurun_out/r04g/fuzz_parity.txt 2>&1; echo "fuzz_parity rc=$?" )
tail -3 gpurun_out/r04g/fuzz_parity.txt | cut -c1-300
grep -c "^adj" gpurun_out/r04g/fuzz_parity.txt; grep "^BAD" -A3 gpurun_out/r04g/fuzz_parity.txt | head -30 | cut -c1-400
grep -c "tall_team': -64" gpurun_out/r04g/fuzz_parity.txt
