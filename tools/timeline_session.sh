#!/bin/bash
# kernel timeline of one pack + solve step: bash tools/timeline_session.sh <out> <workload> <rank> <ranks>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
O=gpurun_out/${1:-tl}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python tools/share_trace.py ${2:-ml20m_user} ${3:-0} 5 ${4:-1} > $O/run.txt 2>&1
grep "^step" $O/run.txt
python tools/share_timeline.py $O/trace > $O/timeline.txt 2>&1
head -70 $O/timeline.txt | cut -c1-170
find $O -name "*.csv" -size +20M -delete
