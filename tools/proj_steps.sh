#!/bin/bash
# the 8-share projection of MovieLens-20M, every timed step of every share, per value of an environment switch (A/B of small-batch latency)
#   bash tools/proj_steps.sh <out-dir> NAME=v1,v2 [reps]
cd "$(dirname "$0")/.."
out=$1; var=${2%%=*}; vals=$(echo ${2#*=} | tr , ' '); reps=${3:-2}
mkdir -p $out
for rep in $(seq 1 $reps); do for v in $vals; do
  env $var=$v GDMIX_BENCH_LINE=full timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-alone --no-cli --c5-full-entities 0 --strong-steps 6 \
      --detail-file $out/${var}_${v}_$rep.json > /dev/null 2> $out/err.txt || tail -3 $out/err.txt
  python - $out/${var}_${v}_$rep.json "$var=$v rep=$rep" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for p in d["detail"].get("strong_projection") or []:
    steps = [[round(x, 2) for x in r.get("step_wall_ms", [])] for r in p["per_rank"]]
    print(sys.argv[2], p["workload"], "ms", round(p["ms"], 3), "mean", round(p.get("ms_mean", 0), 3), "steps", steps)
PY
done; done
