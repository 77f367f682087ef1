#!/bin/bash
mkdir -p gpurun_out/r04j
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "zipf_and_movielens" > gpurun_out/r04j/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r04j/tests.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fe --project-ranks 0 > gpurun_out/r04j/bench.json 2> gpurun_out/r04j/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04j/bench.json') if l.startswith('{')][0])
det = d['detail']
print('value', d['value'], 'handover', det['host_handover']['entities_per_s'], 'serial', det['host_handover']['serial_one_stream']['ms'])
print('cli c2', det['cli_end_to_end']['cold_entities_per_s'], det['cli_end_to_end']['warm_start_entities_per_s'], 'sub', det['cli_subprocess']['cold_s'], det['cli_subprocess']['warm_start_s'])
for k in ('cli_end_to_end_c5', 'cli_end_to_end_ml20m_movie'):
    v = det[k]; print(k, {a: b for a, b in v.items() if a != 'what'})
PY
