#!/bin/bash
export GDMIX_BENCH_LINE=full   # these scripts read the full result from stdout (bench.py prints the short line otherwise)
# round-3 evidence session (GPU box): profiles of every bench workload, the Zipf tail's traffic, the host path, the fixed effect
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03ev
bash tools/profile_round.sh > gpurun_out/r03ev/profile_round.log 2>&1
cp gpurun_out/prof_final/summary.txt gpurun_out/r03ev/c2_summary.txt; cp gpurun_out/prof_final/latest_traffic.json gpurun_out/r03ev/
for w in ml20m_user ml20m_movie c5share; do
  timeout 1500 bash tools/profile_workload.sh $w $w pmc > gpurun_out/r03ev/prof_$w.log 2>&1
  cp gpurun_out/prof_$w/summary.txt gpurun_out/r03ev/${w}_summary.txt; cp gpurun_out/prof_$w/bench_line.json gpurun_out/r03ev/${w}_bench_line.json
done
timeout 1500 bash tools/zipf_pmc.sh 1000000 > gpurun_out/r03ev/zipf_pmc.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --workload zipf --entities 1000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli > gpurun_out/r03ev/zipf_bench.json 2> gpurun_out/r03ev/zipf_bench.err
( echo "== tools/io_parts.py 1000000"; PYTHONPATH=. timeout 900 python tools/io_parts.py 1000000; echo "== tools/e2e_bench.py"; PYTHONPATH=. timeout 1500 python tools/e2e_bench.py 1000000 8 ) > gpurun_out/r03ev/host_path.txt 2>&1
for m in uniform zipf; do FE_BENCH_PATHS=stepping PYTHONPATH=. timeout 600 python tools/fe_bench.py 4000000 32 100000 $m 2>&1 | tail -1; done > gpurun_out/r03ev/fe_bench.txt
ls -la gpurun_out/r03ev
