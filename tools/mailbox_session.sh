#!/bin/bash
# device -> host hand-over without a stream synchronise (GDMIX_RE_MAILBOX=0/1): parity first, then the workloads and the 8-share projection
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=.:tests
out=gpurun_out/${1:-mailbox}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
timeout 1500 python tools/ab.py --env GDMIX_RE_MAILBOX=0,1 --workloads ml20m_user,ml20m_movie,c2 --reps 2 --out $out/ab > $out/ab.txt 2>&1
grep -v "^--" $out/ab.txt | tail -16
for rep in 1 2; do for m in 0 1; do
  GDMIX_RE_MAILBOX=$m GDMIX_BENCH_LINE=full timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-alone --c5-full-entities 0 --detail-file $out/p${m}_$rep.json > /dev/null 2> $out/p${m}_$rep.err
  python - <<PY
import json
d=json.load(open("$out/p${m}_$rep.json"))
print("MAILBOX=$m rep=$rep", "c2 ms", round(d["ms_per_step"],3), [(p["workload"], round(p["ms"],3), round(p.get("ms_mean",0),3)) for p in (d["detail"].get("strong_projection") or [])])
PY
done; done
