#!/bin/bash
# fixed effect: tests, then A/B over GDMIX_FE_COMPRESS (bit 0 rows, bit 1 columns) on the uniform and the Zipf shard
O=gpurun_out/$1; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fixed_effect.py tests/test_fe_model.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
cp gdmix_amd/libgdmix_re.so gdmix_amd/lib_cur.so
for rep in 1 2 3; do for c in 0 1 2 3; do GDMIX_FE_COMPRESS=$c bash tools/fe_ab.sh cur 2>&1 | sed "s/^/compress=$c /"; done; done | tee $O/ab.txt
