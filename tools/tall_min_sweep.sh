#!/bin/bash
# From how many samples on does a skinny entity (p <= 64) go to the tall kernels instead of a group class?  (gdmix_re_set_tall_min_n, default 32)
#   bash tools/tall_min_sweep.sh <out-name> "32 64 96 129 257" [reps]
# per value: MovieLens-20M per-user and per-movie as ONE batch (bench.py --workload ml20m_*), and the 8-share projection of both
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export GDMIX_BENCH_LINE=full PYTHONPATH=.:tests
O=gpurun_out/${1:?out}; VALS=${2:-"32 64 129"}; REPS=${3:-2}
rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-e2e --no-fe --no-alone --no-cli --no-other-workloads --c5-full-entities 0"
for rep in $(seq 1 $REPS); do for v in $VALS; do
  for w in ml20m_user ml20m_movie; do
    timeout 600 python bench.py --steps 10 --warmup 3 --workload $w --tall-min-n $v --project-ranks 0 $COMMON --detail-file $O/${w}_${v}_$rep.json > /dev/null 2> $O/err.txt || tail -3 $O/err.txt
  done
  timeout 900 python bench.py --steps 3 --warmup 2 --tall-min-n $v --strong-steps 6 $COMMON --detail-file $O/proj_${v}_$rep.json > /dev/null 2> $O/err.txt || tail -3 $O/err.txt
  python - $O $v $rep <<'PY'
import json, sys
o, v, rep = sys.argv[1:4]
out = [f"tall_min_n={v} rep={rep}"]
for w in ("ml20m_user", "ml20m_movie"):
    try:
        d = json.load(open(f"{o}/{w}_{v}_{rep}.json"))
        out.append(f"{w} whole {d['ms_per_step']:.3f} ms (solve {d['detail'].get('solve_ms_per_step', float('nan')):.3f})")
    except Exception as e:
        out.append(f"{w}: {e!r}")
try:
    d = json.load(open(f"{o}/proj_{v}_{rep}.json"))
    for p in d["detail"].get("strong_projection") or []:
        out.append(f"{p['workload']} shares max {p['ms']:.3f} mean {p.get('ms_mean', 0):.3f} ms")
except Exception as e:
    out.append(f"proj: {e!r}")
print(" | ".join(out))
PY
done; done | tee $O/summary.txt
