#!/bin/bash
# kernel-trace stats of the Zipf workload (pack + solve), top kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
E=${1:-1000000}
O=gpurun_out/zipf_stats
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python bench.py --workload zipf --entities $E --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/stats.log 2>&1
python tools/prof_summary.py --stats $(ls $O/stats/*.db | head -1) 2>&1 | head -45
rm -f $O/*/*.db
