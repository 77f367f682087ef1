#!/bin/bash
export GDMIX_BENCH_LINE=full   # these scripts read the full result from stdout (bench.py prints the short line otherwise)
# rocprofv3 evidence for one bench workload (run on the GPU box):  tools/profile_workload.sh <workload> <out-name> [pmc]
# kernel-trace stats always; with "pmc" also the FETCH_SIZE / WRITE_SIZE / SQ passes, each in its own run.
W=${1:-c2}; NAME=${2:-$W}; PMC=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads"
O=gpurun_out/prof_$NAME
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
S=$(ls $O/stats/*.db | head -1)
if [ "$PMC" = "pmc" ]; then
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -o q -- $CMD > $O/sq.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/sq2 -o q2 -- $CMD > $O/sq2.log 2>&1
  F=$(ls $O/fetch/*.db | head -1); Wd=$(ls $O/write/*.db | head -1); Q=$(ls $O/sq/*.db | head -1); Q2=$(ls $O/sq2/*.db | head -1)
  python tools/prof_summary.py --stats $S --pmc $F $Wd $Q $Q2 > $O/summary.txt 2>&1
else
  python tools/prof_summary.py --stats $S > $O/summary.txt 2>&1
fi
grep "^{" $O/stats.log > $O/bench_line.json
find $O -name "*.db" -delete
head -40 $O/summary.txt
