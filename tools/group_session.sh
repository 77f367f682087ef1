#!/bin/bash
# partitions per device batch on the GPU box: the model tests, then the C5-shaped and C2 CLI legs with and without grouping
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=.:tests
out=gpurun_out/${1:-group}
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_chain.py -x -q -m gpu > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt
tail -15 $out/tests.txt
for rep in 1 2; do for k in 1 8; do
  GDMIX_PARTITIONS_PER_BATCH=$k GDMIX_BENCH_LINE=full timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-alone --project-ranks 0 --c5-full-entities 0 --detail-file $out/k${k}_$rep.json > $out/k${k}_$rep.out 2> $out/k${k}_$rep.err
  python - <<PY
import json
d=json.load(open("$out/k${k}_$rep.json"))
dd=d["detail"]
g=lambda *p: __import__("functools").reduce(lambda o,k: (o or {}).get(k) if isinstance(o,dict) else None, p, dd)
print("K=$k rep=$rep", "cold", g("cli_end_to_end","cold_entities_per_s"), "warm", g("cli_end_to_end","warm_start_entities_per_s"), "child_s", g("cli_subprocess","cold_s"), "c5", g("cli_end_to_end_c5","entities_per_s"), "movie", g("cli_end_to_end_ml20m_movie","entities_per_s"))
print("   ", {k: (v if not isinstance(v,(dict,list)) else "...") for k,v in (g("cli_end_to_end_c5") or {}).items()})
PY
done; done
