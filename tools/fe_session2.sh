#!/bin/bash
O=gpurun_out/$1; shift; mkdir -p $O
BUILDS="$*"
cd $GRAFT_REPO_ROOT
cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so
for rep in 1 2; do for v in $BUILDS; do DISTS=uniform bash tools/fe_ab.sh $v 2>&1; done; done | tee $O/ab.txt
cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so
timeout 1500 python -m pytest tests/test_fixed_effect.py tests/test_gpu_chain.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/tests.log | cut -c1-400
PYTHONPATH=. timeout 600 python tools/chain_demo.py 2>&1 | grep -v "^INFO" | tail -12
