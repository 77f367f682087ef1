#!/bin/bash
# round 5, evidence session on the GPU box: suite + smoke, rocprofv3 evidence of the C2 bench command (-> profiles/r05_final_c2_1m.txt,
# latest_traffic.json), the fixed effect under rocprofv3 (uniform + Zipf: -> profiles/r05_fe_counters.txt), the default bench line
O=gpurun_out/r05ev; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh > $O/profile_round.txt 2>&1
cp gpurun_out/prof_final/summary.txt $O/c2_summary.txt; cp gpurun_out/prof_final/summary_alone.txt $O/c2_summary_alone.txt; cp gpurun_out/prof_final/latest_traffic.json $O/latest_traffic.json
for m in uniform zipf; do timeout 1200 bash tools/fe_prof_args.sh gpurun_out/fe_$m 4000000 32 100000 $m > $O/fe_$m.log 2>&1; cp gpurun_out/fe_$m/summary.txt $O/fe_${m}_summary.txt; done
cp $O/latest_traffic.json profiles/latest_traffic.json
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/bench_detail.json > $O/bench.out 2> $O/bench.err
echo "bench rc=$? stdout bytes $(wc -c < $O/bench.out) lines $(wc -l < $O/bench.out)"
cat $O/bench.out
