#!/bin/bash
# selected GPU tests + optional commands:  gpurun -- bash tools/session_tests.sh <out-name> "<pytest args>" ["<command>" ...]
O=gpurun_out/$1; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest $2 -m gpu -x -q -s > $O/tests.log 2>&1
echo "tests rc=$?"; grep -v "^INFO\|^DEBUG" $O/tests.log | tail -25 | cut -c1-300
shift 2
i=0
for c in "$@"; do i=$((i+1)); echo "== $c"; bash -c "$c" > $O/cmd$i.log 2>&1; echo "rc=$?"; grep -v "^INFO" $O/cmd$i.log | tail -30 | cut -c1-400; done
