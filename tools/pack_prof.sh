#!/bin/bash
# kernel trace + wait counters of the pack kernels on one workload: bash tools/pack_prof.sh <out-name> [workload]
export GDMIX_BENCH_LINE=full
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${2:-c2}
CMD="python bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --no-alone --project-ranks 0"
O=gpurun_out/${1:-packprof}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -o q -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
S=$(ls $O/stats/*.db | head -1); Q=$(ls $O/sq/*.db | head -1); F=$(ls $O/fetch/*.db | head -1); Wd=$(ls $O/write/*.db | head -1)
python tools/prof_summary.py --stats $S --pmc $F $Wd $Q > $O/summary.txt 2>&1
grep -i "pack\|scan\|kernel  " $O/summary.txt | head -60
find $O -name "*.db" -delete
