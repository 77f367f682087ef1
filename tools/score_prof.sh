#!/bin/bash
# rocprofv3 evidence for the scoring pass: kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in their own passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
SHAPE=${1:-c2}; E=${2:-1000000}
O=gpurun_out/prof_score
rm -rf $O; mkdir -p $O
CMD="python tools/score_bench.py $SHAPE $E"
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -o q -- $CMD > $O/sq.log 2>&1
{ echo "# $CMD under rocprofv3: --kernel-trace --stats, then separate --pmc passes (tools/score_prof.sh)"; grep "ms per pass" $O/stats.log | sed 's/^/# /';
  python tools/prof_summary.py --stats $(ls $O/stats/*.db | head -1) --pmc $(ls $O/fetch/*.db | head -1) $(ls $O/write/*.db | head -1) $(ls $O/sq/*.db | head -1); } > $O/summary_${SHAPE}.txt 2>&1
grep -E "re_score|ms per pass" $O/summary_${SHAPE}.txt | head -20
find $O -name "*.db" -delete
