#!/bin/bash
# ONE parametrised session for the GPU box (replaces round 5's one-off *_session.sh scripts; `git log -- tools/` has them):
#   gpurun --timeout 3000 -- 'bash tools/gpu_session.sh <out-name> step=value [step=value ...]'
# Steps run in the order given, outputs under gpurun_out/<out-name>/, a short tail of each on stdout:
#   tests="<pytest args>"        GPU tests first (-m gpu -x -q), e.g. tests="tests/test_gpu_parity.py tests/test_gpu_fuzz.py"
#   lib=<path.so>                copy this build over gdmix_amd/libgdmix_re.so for the steps that follow
#   ab="<tools/ab.py args>"      e.g. ab="--lib gdmix_amd/lib_a.so,gdmix_amd/lib_b.so --workloads c2,ml20m_user --reps 2"
#   project="NAME=v1,v2" | project="lib:a.so,b.so"   the 8-share projection (and the C2 step) per environment value / build, two runs each
#   cli="NAME=v1,v2"             bench.py's CLI legs (C2 cold / warm, child process, C5-shaped, MovieLens per movie) per environment value
#   fe="<build> [<build> ...]"   fixed-effect bench (tools/fe_ab.sh) of gdmix_amd/lib_<build>.so, three runs each; FE_ENV="NAME=v1,v2" crosses an environment switch
#   timeline="<workload> [rank [ranks]]"   kernel timeline of one pack + solve step (tools/timeline_session.sh)
#   prof="<workload>"            kernel-trace stats + wait / traffic counters of the step's kernels (tools/pack_prof.sh)
#   cmd="<shell command>"        anything else
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=.:tests
name=${1:?out-name}; shift
out=gpurun_out/$name; mkdir -p $out
cp gdmix_amd/libgdmix_re.so $out/lib_at_start.so
n=0
bench_once() {   # $1 = label, rest = extra env assignments; prints the C2 step, the projection and the CLI legs that ran
  local label=$1; shift
  env "$@" GDMIX_BENCH_LINE=full timeout 1200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-fe --no-alone --c5-full-entities 0 $BENCH_EXTRA --detail-file $out/b.json > /dev/null 2> $out/b.err
  python - "$label" $out/b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); dd = d["detail"]
g = lambda *p: __import__("functools").reduce(lambda o, k: o.get(k) if isinstance(o, dict) else None, p, dd)
proj = [(p["workload"], round(p["ms"], 3), round(p.get("ms_mean", 0), 3)) for p in (dd.get("strong_projection") or [])]
cli = {k: v for k, v in (("cold", g("cli_end_to_end", "cold_entities_per_s")), ("warm", g("cli_end_to_end", "warm_start_entities_per_s")),
                         ("child_s", g("cli_subprocess", "cold_s")), ("c5", g("cli_end_to_end_c5", "entities_per_s")),
                         ("movie", g("cli_end_to_end_ml20m_movie", "entities_per_s"))) if v}
print(sys.argv[1], "c2 ms", round(d["ms_per_step"], 3), "proj8", proj, "cli", cli)
PY
}
for step in "$@"; do
  n=$((n + 1)); key=${step%%=*}; val=${step#*=}; log=$out/${n}_$key.txt
  echo "== $key: $val"
  case $key in
    tests) timeout 3000 python -m pytest $val -m gpu -x -q > $log 2>&1; echo "tests rc=$?" >> $log; grep -v "^INFO\|^DEBUG" $log | tail -4 | cut -c1-300 ;;
    lib) cp $val gdmix_amd/libgdmix_re.so; export GDMIX_ALLOW_STALE_LIB=1 ;;   # builds of other sources: the loader refuses those otherwise
    ab) timeout 2400 python tools/ab.py $val --out $out/ab$n > $log 2>&1; grep -v "^--" $log | tail -24 | cut -c1-260 ;;
    project|cli)
      BENCH_EXTRA=$([ $key = project ] && echo "--no-cli" || echo "--project-ranks 0")
      for rep in 1 2; do
        if [ "${val%%:*}" = lib ]; then export GDMIX_ALLOW_STALE_LIB=1; for L in $(echo ${val#lib:} | tr , ' '); do cp $L gdmix_amd/libgdmix_re.so; bench_once "$L rep=$rep"; done
        else for v in $(echo ${val#*=} | tr , ' '); do bench_once "${val%%=*}=$v rep=$rep" "${val%%=*}=$v"; done; fi
      done 2>&1 | tee $log | cut -c1-400 ;;
    fe) cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so; export GDMIX_ALLOW_STALE_LIB=1
        for rep in 1 2 3; do for b in $val; do
          if [ -n "${FE_ENV:-}" ]; then for v in $(echo ${FE_ENV#*=} | tr , ' '); do env ${FE_ENV%%=*}=$v bash tools/fe_ab.sh $b 2>&1 | sed "s/^/${FE_ENV%%=*}=$v /"; done
          else bash tools/fe_ab.sh $b 2>&1; fi
        done; done | tee $log; cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so ;;
    timeline) bash tools/timeline_session.sh $name/tl$n $val 2>&1 | tee $log | sed -n 1,45p | cut -c1-160 ;;
    prof) bash tools/pack_prof.sh $name/prof$n $val > $log 2>&1; head -32 $out/prof$n/summary.txt | cut -c1-150 ;;
    cmd) bash -c "$val" > $log 2>&1; echo "rc=$?"; grep -v "^INFO" $log | tail -30 | cut -c1-400 ;;
    *) echo "unknown step $key" ;;
  esac
done
cp $out/lib_at_start.so gdmix_amd/libgdmix_re.so; rm -f $out/lib_at_start.so; unset GDMIX_ALLOW_STALE_LIB
