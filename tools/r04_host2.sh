#!/bin/bash
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 100000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for p in d['detail']['strong_projection'][:2]:
    r=p['per_rank'][0]
    print(p['workload'], 'ms', round(p['ms'],2), 'rank0: wall', round(r['ms_per_step'],2), 'pack', round(r['pack_ms'],2), r['largest_launches'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r04h/prof -o t -- python bench.py --scaling strong --workload ml20m_user --gpus 1 --steps 5 --warmup 2 --no-rebalance > gpurun_out/r04h/strong1.log 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('gpurun_out/r04h/prof/**/*.db', recursive=True)
print(db)
PY
ls gpurun_out/r04h/prof/* | head; find gpurun_out/r04h -name "*stats*" | head
