#!/bin/bash
# A/B of two builds on the MovieLens workloads + the 8-share projection: bash tools/pack_session2.sh <out> <libA> <libB>
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=.:tests
out=gpurun_out/${1:-pack2}; A=$2; B=$3
mkdir -p $out
cp $B gdmix_amd/libgdmix_re.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
bash tools/timeline_session.sh ${1:-pack2}/tl ml20m_user 0 1 | sed -n 6,32p | cut -c1-150
timeout 1500 python tools/ab.py --lib $A,$B --workloads ml20m_user,ml20m_movie,c5share --reps 2 --steps 6 --warmup 2 --out $out/ab > $out/ab.txt 2>&1
grep -v "^--" $out/ab.txt | tail -18
for rep in 1 2; do for L in $A $B; do
  cp $L gdmix_amd/libgdmix_re.so
  GDMIX_BENCH_LINE=full timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-alone --c5-full-entities 0 --detail-file $out/p.json > /dev/null 2> $out/p.err
  python - <<PY
import json
d=json.load(open("$out/p.json"))
print("$L rep=$rep", "c2 ms", round(d["ms_per_step"],3), [(p["workload"], round(p["ms"],3), round(p.get("ms_mean",0),3)) for p in (d["detail"].get("strong_projection") or [])])
PY
done; done
cp $B gdmix_amd/libgdmix_re.so
