#!/bin/bash
mkdir -p gpurun_out/r03h
cd /root/repo
timeout 1500 python -m pytest tests/test_fixed_effect.py tests/test_fe_model.py -m gpu -q -x 2>&1 | tail -3
for m in uniform zipf; do PYTHONPATH=. python tools/fe_bench.py 4000000 32 100000 $m 2>&1 | tail -1; done > gpurun_out/r03h/fe_bench.txt
python - <<'PY'
import json
for l in open("gpurun_out/r03h/fe_bench.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["columns"], {k: v for k, v in d.items() if not isinstance(v, (dict, list))})
PY
