#!/bin/bash
# fixed effect, round 5: tests of the touched paths, then A/B of builds (gdmix_amd/lib_<name>.so) x one-launch step x look-ahead
#   gpurun -- bash tools/fe_session.sh <out-name> <build> [<build> ...]
O=gpurun_out/$1; shift; mkdir -p $O
BUILDS="$*"
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fixed_effect.py tests/test_fe_model.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/tests.log | cut -c1-300
for rep in 1 2; do
for v in $BUILDS; do
  for cfg in "0 0" "1 0" "1 2"; do
    f=${cfg% *}; a=${cfg#* }
    GDMIX_FE_FUSED_TAIL=$f GDMIX_FE_LOOKAHEAD=$a bash tools/fe_ab.sh $v 2>&1 | sed "s/^/fused=$f ahead=$a /"
  done
done
done | tee $O/ab.txt
