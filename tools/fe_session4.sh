#!/bin/bash
# fixed effect: tests, A/B of builds, then the rocprofv3 evidence of the final library (uniform + Zipf)
O=gpurun_out/$1; shift; mkdir -p $O
BUILDS="$*"
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fixed_effect.py tests/test_fe_model.py tests/test_gpu_chain.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in $BUILDS; do DISTS=uniform bash tools/fe_ab.sh $v 2>&1; done; done | tee $O/ab.txt
cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so
for m in uniform zipf; do timeout 1200 bash tools/fe_prof_args.sh gpurun_out/fe_$m 4000000 32 100000 $m > $O/fe_$m.log 2>&1; cp gpurun_out/fe_$m/summary.txt $O/fe_${m}_summary.txt; done
grep -E "fe_tail|fe_scatter|fe_finish" $O/fe_uniform_summary.txt | head -8
