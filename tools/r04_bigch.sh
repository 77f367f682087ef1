#!/bin/bash
# the big-entity pack's chunk size (BIG_CH in csrc/re_pack_big.hip; rebuilt per value): pack ms of the MovieLens / Zipf workloads and the 8-share projection
mkdir -p gpurun_out/tt
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pack" 2>&1 | tail -2
for w in ml20m_movie ml20m_user zipf; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --project-ranks 0 --no-alone > gpurun_out/tt/ch_${w}.json 2> gpurun_out/tt/ch_${w}.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/tt/ch_${w}.json') if l.startswith('{')][0])
    det = d['detail']
    print('$w', 'ms/step', round(d['ms_per_step'], 3), 'pack', round(det['pack_ms_per_step'], 3), 'solve', round(det['solve_ms_per_step'], 3))
except Exception as e:
    print('$w failed', e, open('gpurun_out/tt/ch_${w}.err').read()[-400:])
PY
done
timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 100000 --strong-steps 5 --no-alone > gpurun_out/tt/ch_proj.json 2> gpurun_out/tt/ch_proj.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/tt/ch_proj.json') if l.startswith('{')][0])
for p in d['detail'].get('strong_projection') or []:
    if p['workload'] != 'c5':
        print('projection', p['workload'], 'ms', round(p['ms'], 3), 'pack', [round(r['pack_ms'], 2) for r in p['per_rank']], [round(r['ms_per_step'], 2) for r in p['per_rank']])
PY
