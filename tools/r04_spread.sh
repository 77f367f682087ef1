#!/bin/bash
# exploration: large classes on different streams (GDMIX_RE_SPREAD = queues, GDMIX_RE_SPREAD_ORDER=1: most lanes first)
mkdir -p gpurun_out/tt
cd /root/repo
for cfg in "0 0" "2 0" "3 0" "4 0" "2 1" "3 1" "4 1" "0 0"; do
  set -- $cfg
  for w in c2 c5share; do
    GDMIX_RE_SPREAD=$1 GDMIX_RE_SPREAD_ORDER=$2 timeout 600 python bench.py --workload $w --c5-entities 1000000 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --project-ranks 0 > gpurun_out/tt/sp_${w}_$1_$2.json 2> gpurun_out/tt/sp_${w}_$1_$2.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/tt/sp_${w}_$1_$2.json') if l.startswith('{')][0])
    det = d['detail']
    print('$w queues $1 order $2', 'ms/step', round(d['ms_per_step'], 3), 'pack', round(det['pack_ms_per_step'], 3), 'solve', round(det['solve_ms_per_step'], 3))
except Exception as e:
    print('$w $cfg failed', e, open('gpurun_out/tt/sp_${w}_$1_$2.err').read()[-300:])
PY
  done
done
