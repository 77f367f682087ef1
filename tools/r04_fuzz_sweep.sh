#!/bin/bash
# randomised parity sweeps on the round-4 library (seeds disjoint from rounds 1-3)
mkdir -p gpurun_out/r04f
export PYTHONPATH=.:tests
( python tools/fuzz_parity.py 4000 1000000 > gpurun_out/r04f/fuzz_parity.txt 2>&1; echo "fuzz_parity rc=$?" ) 
tail -3 gpurun_out/r04f/fuzz_parity.txt | cut -c1-300
( python tools/fuzz_fe.py 300 1100000 > gpurun_out/r04f/fuzz_fe.txt 2>&1; echo "fuzz_fe rc=$?" )
tail -2 gpurun_out/r04f/fuzz_fe.txt | cut -c1-300
( GDMIX_FE_HOT_MIN=300 GDMIX_FE_WINDOW_BITS=11 GDMIX_FE_CHUNK=3001 python tools/fuzz_fe.py 80 1200000 > gpurun_out/r04f/fuzz_fe_cut.txt 2>&1; echo "fuzz_fe cut rc=$?" )
tail -2 gpurun_out/r04f/fuzz_fe_cut.txt | cut -c1-300
grep -c "^adj" gpurun_out/r04f/fuzz_parity.txt; grep "^BAD" -A3 gpurun_out/r04f/fuzz_parity.txt | head -20 | cut -c1-400
