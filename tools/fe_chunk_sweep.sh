#!/bin/bash
# fixed effect: column / row pass time against the unit length (GDMIX_FE_CHUNK), uniform and Zipf columns
mkdir -p gpurun_out/r03h
cd /root/repo
: > gpurun_out/r03h/fe_sweep.txt
for c in 8192 16384 32768 62500 125000; do for m in uniform zipf; do
  echo "chunk $c $m" >> gpurun_out/r03h/fe_sweep.txt
  GDMIX_FE_CHUNK=$c FE_BENCH_PATHS=stepping PYTHONPATH=. python tools/fe_bench.py 4000000 32 100000 $m 2>&1 | tail -1 >> gpurun_out/r03h/fe_sweep.txt
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r03h/fe_sweep.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("   ms/eval %.3f frac %.3f rows %.3f cols %.3f" % (d["ms_per_evaluation"], d["frac"], d["rows_pass_ms"], d["cols_pass_ms"]))
    else: print(l.strip())
PY
