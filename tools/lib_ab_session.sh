#!/bin/bash
# A/B of builds of libgdmix_re.so on one workload, with what the step time alone does not show (round 6):
#   bash tools/lib_ab_session.sh <out-name> <workload> "<lib names>" [pmc-lib names]
# per build gdmix_amd/lib_<name>.so (or `cur` = the library in place):
#   1. every solve kernel's duration ALONE (GDMIX_RE_SPREAD=0, classes one after another) from a rocprofv3 kernel trace
#   2. for the builds listed in [pmc-lib names]: the LDS counters of the solve kernels (SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS ...)
# then tools/ab.py over all builds for the step itself (three rounds, minimum and every run). Summary: gpurun_out/<out-name>/summary.txt
export GDMIX_BENCH_LINE=full GDMIX_ALLOW_STALE_LIB=1 PYTHONPATH=.:tests
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:?out}; W=${2:-c2}; LIBS=${3:-cur}; PMC=${4:-}
rm -rf $O; mkdir -p $O
cp gdmix_amd/libgdmix_re.so $O/lib_cur_keep.so; cp gdmix_amd/libgdmix_re.so gdmix_amd/lib_cur.so
CMD="python bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --no-alone --project-ranks 0"
for L in $LIBS; do
  cp gdmix_amd/lib_$L.so gdmix_amd/libgdmix_re.so
  GDMIX_RE_SPREAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/st_$L -o s -- $CMD > $O/st_$L.log 2>&1
  S=$(ls $O/st_$L/*.db 2>/dev/null | head -1)
  { echo "== $L: kernels alone (GDMIX_RE_SPREAD=0)"; [ -n "$S" ] && python tools/prof_summary.py --stats $S | grep -v "^gdmix::\|at::native\|rocclr" | sed -n 2,14p; } >> $O/summary.txt
  for P in $PMC; do if [ "$P" = "$L" ]; then
    GDMIX_RE_SPREAD=0 timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES -d $O/q_$L -o q -- $CMD > $O/q_$L.log 2>&1
    Q=$(ls $O/q_$L/*.db 2>/dev/null | head -1)
    { echo "== $L: LDS counters per solve kernel (sum over dispatches)"; [ -n "$Q" ] && python tools/prof_summary.py --pmc $Q | grep "re_solve" | cut -c1-130 | sort; } >> $O/summary.txt
  fi; done
  find $O -name "*.db" -delete
done
cp $O/lib_cur_keep.so gdmix_amd/libgdmix_re.so
LIST=$(for L in $LIBS; do printf "gdmix_amd/lib_%s.so," $L; done); LIST=${LIST%,}
{ echo "== step (tools/ab.py, 3 rounds)"; timeout 2400 python tools/ab.py --lib $LIST --workloads $W --reps 3 --out $O/ab 2>&1 | grep -v "^--"; } >> $O/summary.txt
cp $O/lib_cur_keep.so gdmix_amd/libgdmix_re.so; rm -f $O/lib_cur_keep.so gdmix_amd/lib_cur.so
cat $O/summary.txt | cut -c1-220
