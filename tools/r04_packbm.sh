#!/bin/bash
# the bitmap path of the pack (entities of at most 128 non-zeros with columns below 2 048): parity tests, then A/B on C2 and the MovieLens workloads
mkdir -p gpurun_out/tt
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "default_routing_matches or pack" 2>&1 | tail -4
for t in 0 1 0 1; do
  for w in c2 ml20m_user; do
    GDMIX_PACK_BITMAP=$t timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --project-ranks 0 --no-alone > gpurun_out/tt/bm_${w}_$t.json 2> gpurun_out/tt/bm_${w}_$t.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/tt/bm_${w}_$t.json') if l.startswith('{')][0])
    det = d['detail']
    print('$w bitmap $t', 'ms/step', round(d['ms_per_step'], 3), 'pack', round(det['pack_ms_per_step'], 3), 'solve', round(det['solve_ms_per_step'], 3))
except Exception as e:
    print('$w bitmap $t failed', e, open('gpurun_out/tt/bm_${w}_$t.err').read()[-400:])
PY
  done
done
