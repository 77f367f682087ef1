#!/bin/bash
# round 4: how many entities the team class should take (GDMIX_RE_TALL_TEAM_LIMIT) — 8-share projection of the MovieLens populations
mkdir -p gpurun_out/tt
cd /root/repo
for lim in ${LIMITS:-0 16 32}; do
  GDMIX_RE_TALL_TEAM=$([ $lim = 0 ] && echo 0 || echo 1) GDMIX_RE_TALL_TEAM_LIMIT=$lim timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 100000 --strong-steps 5 > gpurun_out/tt/proj_l$lim.json 2> gpurun_out/tt/proj_l$lim.err
  echo "limit=$lim rc=$?"
done
python - <<'PY'
import json
import os
for lim in [int(x) for x in os.environ.get('LIMITS', '0 16 32').split()]:
    try:
        d = json.loads([l for l in open(f'gpurun_out/tt/proj_l{lim}.json') if l.startswith('{')][0])
    except Exception as e:
        print('parse failed', lim, e); continue
    for p in d['detail'].get('strong_projection') or []:
        if p['workload'] != 'c5':
            print('limit', lim, p['workload'], 'ms', round(p['ms'], 3), [round(r['ms_per_step'], 2) for r in p['per_rank']])
            worst = max(p['per_rank'], key=lambda r: r['ms_per_step'])
            print('      slowest share', worst['rank'], 'steps', worst.get('step_wall_ms'), [(l['kernel'].replace('re_solve_', '')[:22], l['entities'], l['ms']) for l in worst['largest_launches'][:4]])
    w = d['detail']['workloads']
    print('   1 GPU:', {k: round(v['ms_per_step'], 3) for k, v in w.items() if 'ms_per_step' in v})
PY
