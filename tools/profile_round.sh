#!/bin/bash
export GDMIX_BENCH_LINE=full   # these scripts read the full result from stdout (bench.py prints the short line otherwise)
# rocprofv3 evidence for the bench command (run on the GPU box): kernel-trace stats, then PMC passes on their own
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --no-alone"
O=gpurun_out/prof_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
# the same with the size classes one after another (GDMIX_RE_SPREAD=0): every kernel's duration alone
GDMIX_RE_SPREAD=0 rocprofv3 --kernel-trace --stats -d $O/stats0 -o s0 -- $CMD > $O/stats0.log 2>&1
S0=$(ls $O/stats0/*.db | head -1)
python tools/prof_summary.py --stats $S0 > $O/summary_alone.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/sq -o q -- $CMD > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/sq2 -o q2 -- $CMD > $O/sq2.log 2>&1
S=$(ls $O/stats/*.db | head -1); F=$(ls $O/fetch/*.db | head -1); W=$(ls $O/write/*.db | head -1); Q=$(ls $O/sq/*.db | head -1); Q2=$(ls $O/sq2/*.db | head -1)
python tools/prof_summary.py --stats $S --pmc $F $W $Q $Q2 > $O/summary.txt 2>&1
python tools/make_traffic_json.py $F $W $O/latest_traffic.json $Q --workload "c2 E=1000000 n~16 k=4 D=1024 survey-generator" > /dev/null 2>&1
tail -3 $O/stats.log | head -2
head -30 $O/summary.txt
find $O -name "*.db" -delete
