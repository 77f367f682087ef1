#!/bin/bash
export PYTHONPATH=.
for flags in "$@"; do
  echo "=== flags: $flags"
  GDMIX_EXTRA_FLAGS="$flags" python -m gdmix_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  python tools/fe_bench.py 2>/dev/null | grep stepping | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('ms_per_evaluation','achieved_GBps','rows_pass_ms','cols_pass_ms','device_loop_ms','nfev')})"
done
