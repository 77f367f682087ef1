#!/bin/bash
mkdir -p gpurun_out/r04m
timeout 900 python -m pytest tests/test_fixed_effect.py tests/test_gpu_parity.py -m gpu -x -q -k "two_workers or team_lds or block_kernel or default_routing" > gpurun_out/r04m/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04m/tests.log | cut -c1-600
bash tools/zipf_pmc.sh 1000000 > gpurun_out/r04m/zipf_pmc.txt 2>&1; tail -5 gpurun_out/r04m/zipf_pmc.txt
bash tools/zipf_quick.sh --no-other-workloads
echo "== fe uniform"; bash tools/fe_prof_args.sh gpurun_out/r04m/fe_uniform 4000000 32 100000 uniform
echo "== fe zipf"; bash tools/fe_prof_args.sh gpurun_out/r04m/fe_zipf 4000000 32 100000 zipf
