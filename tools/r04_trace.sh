#!/bin/bash
# kernel timeline of one share (which class launches overlap): rocprofv3 --kernel-trace of tools/share_trace.py
mkdir -p gpurun_out/trace
cd /root/repo
export TMPDIR=/tmp
for w in ml20m_movie ml20m_user; do
  rm -rf /tmp/tr_$w
  PYTHONPATH=. timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$w -- python tools/share_trace.py $w 3 6 > gpurun_out/trace/$w.log 2>&1
  tail -6 gpurun_out/trace/$w.log
  python tools/share_timeline.py /tmp/tr_$w > gpurun_out/trace/${w}_timeline.txt 2>&1
  tail -60 gpurun_out/trace/${w}_timeline.txt
done
