#!/bin/bash
# HBM traffic of the tail (team) kernels on the Zipf workload: FETCH_SIZE / WRITE_SIZE per dispatch, separate passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
E=${1:-1000000}
CMD="python bench.py --workload zipf --entities $E --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads"
O=gpurun_out/zipf_pmc
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
python - <<PY
import sqlite3, glob
def q(db, sql):
    return sqlite3.connect(glob.glob(db)[0]).cursor().execute(sql).fetchall()
dur = q("$O/stats/*.db", "select name, (end-start)/1e6 from kernels where name like '%re_solve_team%' order by start")
f = q("$O/fetch/*.db", "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%re_solve_team%' group by dispatch_id order by dispatch_id")
w = q("$O/write/*.db", "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like '%re_solve_team%' group by dispatch_id order by dispatch_id")
print(len(dur), len(f), len(w))
for (n, ms), (_, _, fk), (_, _, wk) in zip(dur, f, w):
    print("%-60s %9.3f ms  fetch %9.1f MB  write %9.1f MB  -> %7.1f GB/s" % (n.split("(")[0][-60:], ms, fk/1024, wk/1024, (fk+wk)*1024/ms/1e6))
PY
rm -f $O/*/*.db
