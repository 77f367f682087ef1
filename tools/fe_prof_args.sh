#!/bin/bash
# the fixed effect under rocprofv3 for a given shard shape: bash tools/fe_prof_args.sh <out dir> [fe_bench.py arguments]   e.g.  ... 4000000 32 100000 zipf
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=. FE_BENCH_PATHS=stepping
O=$1; shift
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python tools/fe_bench.py "$@" > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -o q -- python tools/fe_bench.py "$@" > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $O/sq2 -o q2 -- python tools/fe_bench.py "$@" > $O/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- python tools/fe_bench.py "$@" > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- python tools/fe_bench.py "$@" > $O/write.log 2>&1
python tools/prof_summary.py --stats $(ls $O/stats/*.db | head -1) --pmc $(ls $O/sq/*.db | head -1) $(ls $O/sq2/*.db | head -1) $(ls $O/fetch/*.db | head -1) $(ls $O/write/*.db | head -1) > $O/summary.txt 2>&1
grep -E "fe_|rocprim|radix|scan" $O/summary.txt | head -30
tail -3 $O/stats.log
find $O -name "*.db" -delete
