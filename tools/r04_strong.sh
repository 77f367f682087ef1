#!/bin/bash
# round 4, first GPU session: the strong-scaling harness tests, then the default bench line with the 8-rank projection
mkdir -p gpurun_out/r04a
timeout 1500 python -m pytest tests/test_bench_harness.py -m gpu -x -q > gpurun_out/r04a/harness.log 2>&1
echo "harness rc=$?"; tail -15 gpurun_out/r04a/harness.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r04a/bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r04a/bench.json') if l.startswith('{')][0])
    print('value', d['value'], 'ms', d['ms_per_step'])
    for p in d['detail']['strong_projection'] or []:
        print(p['workload'], 'ms', round(p['ms'], 2), 'ent/s', round(p['entities_per_s']), 'imb', round(p['imbalance'], 3),
              [round(r['ms_per_step'], 2) for r in p['per_rank']], p['rebalance_plan']['predicted_imbalance'], p['rebalance_plan']['after_imbalance'],
              p['rebalance_plan']['wire_bytes_to_move'])
    for k, v in (d['detail']['workloads'] or {}).items():
        print(k, {a: b for a, b in v.items() if a in ('ms_per_step', 'entities_per_s', 'skipped')})
    print('handover', d['detail']['host_handover']['entities_per_s'], d['detail']['host_handover']['serial_one_stream'])
    print('cli', d['detail']['cli_end_to_end']['cold_entities_per_s'], d['detail']['cli_end_to_end']['warm_start_entities_per_s'], d['detail']['cli_subprocess'])
except Exception as e:
    print('parse failed', e)
PY
