#!/bin/bash
# The closing evidence session of a round on the GPU box:  gpurun --timeout 5000 -- 'bash tools/evidence_round.sh r06'
#   1. the whole -m gpu suite + smoke()
#   2. rocprofv3 evidence of the C2 bench command (tools/profile_round.sh): kernel trace (timed schedule and classes alone), FETCH / WRITE /
#      SQ counters in separate --pmc passes  -> gpurun_out/<tag>ev/c2_summary.txt, c2_summary_alone.txt, latest_traffic.json
#   3. the fixed effect under rocprofv3, uniform and Zipf columns (tools/fe_prof_args.sh) -> fe_uniform_summary.txt, fe_zipf_summary.txt
#   4. the default bench line, as the driver runs it -> bench.out (the short line), bench_detail.json
# Copy what is to be judged into profiles/ afterwards (profiles/<tag>_final_c2_1m.txt = c2_summary.txt + the geometry section of
# c2_summary_alone.txt; profiles/latest_traffic.json; profiles/<tag>_fe_counters.txt; profiles/<tag>_final_bench_line.json / _detail.json).
# (Rounds 3 - 5 had one script each: `git log -- tools/` has evidence_session.sh, assemble_profiles.sh, evidence_r05.sh.)
T=${1:?round tag, e.g. r06}
O=gpurun_out/${T}ev; mkdir -p $O
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
