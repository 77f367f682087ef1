#!/bin/bash
# round 4: the team class of the tall kernels (four workgroups on one entity) — tests, then A/B of the MovieLens legs and the 8-share projection
mkdir -p gpurun_out/tt
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tall_team or tallest or default_routing_reaches or tall_kernels_give or lowers_the_tall" > gpurun_out/tt/tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/tt/tests.log | cut -c1-300
for t in 0 1; do
  for w in ml20m_movie ml20m_user; do
    GDMIX_RE_TALL_TEAM=$t timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --project-ranks 0 > gpurun_out/tt/${w}_t$t.json 2> gpurun_out/tt/${w}_t$t.err
    echo "$w team=$t rc=$?"
  done
  GDMIX_RE_TALL_TEAM=$t timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 250000 --strong-steps 3 > gpurun_out/tt/proj_t$t.json 2> gpurun_out/tt/proj_t$t.err
  echo "projection team=$t rc=$?"
done
python - <<'PY'
import json
def load(p):
    try:
        return json.loads([l for l in open(p) if l.startswith('{')][0])
    except Exception as e:
        print('parse failed', p, e); return None
for t in (0, 1):
    for w in ('ml20m_movie', 'ml20m_user'):
        d = load(f'gpurun_out/tt/{w}_t{t}.json')
        if d:
            det = d['detail']
            cls = [(n.replace('re_solve_', ''), c, m) for (n, c), m in zip(det['classes'], det['class_ms']) if c]
            print(w, 'team', t, 'ms/step', round(d['ms_per_step'], 3), 'pack', round(det['pack_ms_per_step'], 3), 'solve', round(det['solve_ms_per_step'], 3), 'ent/s', d['value'], cls)
    d = load(f'gpurun_out/tt/proj_t{t}.json')
    if d:
        for p in d['detail'].get('strong_projection') or []:
            print('projection team', t, p['workload'], 'ms', round(p['ms'], 3), 'ent/s', round(p['entities_per_s']), [round(r['ms_per_step'], 2) for r in p['per_rank']])
PY
