#!/bin/bash
# kernel times of gdmix_fe_create + the loop (tools/fe_phases.py) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONPATH=.
O=gpurun_out/prof_fe_create
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python tools/fe_phases.py "$@" > $O/stats.log 2>&1
python tools/prof_summary.py --stats $(ls $O/stats/*.db | head -1) > $O/summary.txt 2>&1
head -40 $O/summary.txt
find $O -name "*.db" -delete
