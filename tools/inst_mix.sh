#!/bin/bash
# What the vector instructions of a workload's kernels ARE (rocprofv3 --pmc, two passes of their own, classes one after another):
#   bash tools/inst_mix.sh <out-name> [workload]     -> gpurun_out/<out-name>/mix.txt
# per kernel: SQ_INSTS_VALU split into fp64 add / mul / fma / transcendental, int32, int64, conversions and the rest (moves, DPP moves,
# compares, selects: what the counters do not name), and the lanes an instruction had active (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 4).
export GDMIX_BENCH_LINE=full GDMIX_RE_SPREAD=0 PYTHONPATH=.:tests
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:?out}; W=${2:-c2}
CMD="python bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --no-alone --project-ranks 0"
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $O/a -o a -- $CMD > $O/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_BRANCH -d $O/b -o b -- $CMD > $O/b.log 2>&1
A=$(ls $O/a/*.db 2>/dev/null | head -1); B=$(ls $O/b/*.db 2>/dev/null | head -1)
python tools/inst_mix.py $A $B > $O/mix.txt 2> $O/mix.err
tail -q -n 2 $O/a.log $O/b.log | cut -c1-300; cat $O/mix.err | tail -5; cat $O/mix.txt | cut -c1-250
find $O -name "*.db" -delete
