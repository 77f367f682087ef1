#!/bin/bash
# The split between the one-wavefront and the eight-wavefront tall kernels chosen per batch (GDMIX_RE_TALL_ADAPT = the most workgroups
# the eight-wavefront class may get by a lower split; 0 = fixed split of 4 096 samples): parity tests, then the 8-share projection of
# the MovieLens-20M populations and the whole populations on one GPU, each way.
mkdir -p gpurun_out/adapt
cd /root/repo
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_model.py tests/test_rebalance.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/adapt/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/adapt/tests.log | cut -c1-300
for n in ${ADAPT_LIST:-128 0 64 192 256}; do
  GDMIX_RE_TALL_ADAPT=$n timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 500000 --strong-steps 4 > gpurun_out/adapt/bench_$n.json 2> gpurun_out/adapt/bench_$n.err
  echo "== adapt limit $n rc=$?"
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads([l for l in open(f'gpurun_out/adapt/bench_{n}.json') if l.startswith('{')][0])
    print('c2 ms', d['ms_per_step'])
    for p in d['detail']['strong_projection'] or []:
        print(' projection', p['workload'], 'ms', round(p['ms'], 2), 'ent/s', round(p['entities_per_s']), [round(r['ms_per_step'], 2) for r in p['per_rank']])
        if p['workload'].startswith('ml'):
            print('    ', [(k['kernel'][:28], k['entities'], round(k['ms'], 2)) for k in p['per_rank'][3].get('largest_launches', [])][:4])
    for k, v in (d['detail']['workloads'] or {}).items():
        print(' workload', k, {a: b for a, b in v.items() if a in ('ms_per_step', 'entities_per_s', 'skipped')})
except Exception as e:
    print('parse failed', e)
PY
done
