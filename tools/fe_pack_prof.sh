#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
O=gpurun_out/prof_fe_pack
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python tools/fe_phases.py > $O/stats.log 2>&1
python tools/prof_summary.py --stats $(ls $O/stats/*.db | head -1) 2>&1 | head -24
tail -7 $O/stats.log
rm -f $O/*/*.db
