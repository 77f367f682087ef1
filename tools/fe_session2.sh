#!/bin/bash
# A/B of builds on the fixed-effect bench (uniform + zipf), then the tests of the touched paths: gpurun -- bash tools/fe_session2.sh <out> <build> ...
O=gpurun_out/$1; shift; mkdir -p $O
BUILDS="$*"
cd $GRAFT_REPO_ROOT
cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in $BUILDS; do bash tools/fe_ab.sh $v 2>&1; done; done | tee $O/ab.txt
cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so
timeout 2400 python -m pytest tests/test_fixed_effect.py tests/test_fe_model.py tests/test_gpu_chain.py tests/test_gpu_parity.py tests/test_rebalance.py tests/test_bench_harness.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-400
