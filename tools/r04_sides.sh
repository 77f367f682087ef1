#!/bin/bash
# Several side streams for the small classes (GDMIX_RE_SIDE_STREAM=N; 1 = round 3's single side stream): parity tests of the routing,
# then the 8-share projection of the MovieLens-20M populations and the single-GPU workloads, each way.
mkdir -p gpurun_out/sides
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/sides/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/sides/tests.log | cut -c1-300
for n in 3 1 3 1; do
  GDMIX_RE_SIDE_STREAM=$n timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 1000000 --strong-steps 4 > gpurun_out/sides/bench_$n.json 2> gpurun_out/sides/bench_$n.err
  echo "== side streams $n rc=$?"
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f'gpurun_out/sides/bench_{n}.json') if l.startswith('{')][0])
    print('c2 ms', d['ms_per_step'])
    for p in d['detail']['strong_projection'] or []:
        print(' projection', p['workload'], 'ms', round(p['ms'], 2), 'ent/s', round(p['entities_per_s']), [round(r['ms_per_step'], 2) for r in p['per_rank']])
    for k, v in (d['detail']['workloads'] or {}).items():
        print(' workload', k, {a: b for a, b in v.items() if a in ('ms_per_step', 'entities_per_s', 'skipped')})
except Exception as e:
    print('parse failed', e)
PY
done
