#!/bin/bash
# the long randomised parity sweeps on the GPU box:  gpurun -- bash tools/fuzz_sweep.sh <out-name> <re cases> <re first seed> <fe cases> <fe first seed>
O=gpurun_out/$1; mkdir -p $O
cd $GRAFT_REPO_ROOT
export PYTHONPATH=.:tests
( echo "PYTHONPATH=.:tests python tools/fuzz_parity.py $2 $3"; timeout 3000 python tools/fuzz_parity.py $2 $3 2>&1 | grep -v "^ok \|amdgpu.ids" ) > $O/fuzz_re.txt
tail -2 $O/fuzz_re.txt
( echo "PYTHONPATH=.:tests python tools/fuzz_fe.py $4 $5"; timeout 2000 python tools/fuzz_fe.py $4 $5 2>&1 | grep -v "^ok \|amdgpu.ids" ) > $O/fuzz_fe.txt
tail -3 $O/fuzz_fe.txt
