#!/bin/bash
# round 4, closing session: rocprofv3 evidence of the C2 bench command (profiles/r04_final_c2_1m.txt, latest_traffic.json), then the
# default bench line as the driver runs it
mkdir -p gpurun_out/r04z
bash tools/profile_round.sh > gpurun_out/r04z/profile_round.txt 2>&1
cp gpurun_out/prof_final/summary.txt gpurun_out/r04z/summary.txt; cp gpurun_out/prof_final/summary_alone.txt gpurun_out/r04z/summary_alone.txt; cp gpurun_out/prof_final/latest_traffic.json gpurun_out/r04z/latest_traffic.json
tail -3 gpurun_out/prof_final/stats.log | cut -c1-300
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04z/bench.json 2> gpurun_out/r04z/bench.err
echo "bench rc=$?"
python tools/bench_summary.py gpurun_out/r04z/bench.json 2>/dev/null | head -40
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04z/bench.json') if l.startswith('{')][0])
det = d['detail']
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('achieved', 'frac', 'traffic', 'kernel', 'avg_launch_ms')})
print('cpu', d['cpu_baseline'])
for p in det['strong_projection'] or []:
    print(p['workload'], 'ms', round(p['ms'], 2), 'ent/s', round(p['entities_per_s']), 'imb', round(p['imbalance'], 3), 'plan', p['rebalance_plan']['predicted_imbalance'], '->', p['rebalance_plan']['after_imbalance'], p['rebalance_plan']['wire_bytes_to_move'])
for k, v in (det['workloads'] or {}).items():
    print(k, {a: b for a, b in v.items() if a in ('ms_per_step', 'entities_per_s', 'skipped')})
h = det['host_handover']; print('handover', h['entities_per_s'], h['with_feature_index']['entities_per_s'], h['serial_one_stream'])
print('score', det['score_pass']['frac_of_hbm_peak'], 'fe', det['fixed_effect_eval']['ms_per_evaluation'], det['fixed_effect_eval']['frac_of_hbm_peak'])
print('cli', det['cli_end_to_end']['cold_entities_per_s'], det['cli_end_to_end']['warm_start_entities_per_s'], det['cli_subprocess'])
for k in ('cli_end_to_end_c5', 'cli_end_to_end_ml20m_movie'):
    v = det[k]; print(k, {a: b for a, b in v.items() if a in ('entities', 'cold_s', 'entities_per_s', 'phases_thread_s', 'dominant_phase')})
PY
