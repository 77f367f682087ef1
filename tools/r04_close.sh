#!/bin/bash
# closing session of round 4: whole GPU suite, smoke, the default bench line, the projection report and one share's timeline
mkdir -p gpurun_out/r04y gpurun_out/trace
cd /root/repo
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04y/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04y/tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > gpurun_out/r04y/bench.json 2> gpurun_out/r04y/bench.err ) 2>&1 | grep real
python tools/projection_report.py gpurun_out/r04y/bench.json > gpurun_out/r04y/projection.txt 2>&1
grep -E "^==|job =|^#" gpurun_out/r04y/projection.txt | cut -c1-260
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04y/bench.json') if l.startswith('{')][0])
det = d['detail']
print('value', d['value'], 'ms', d['ms_per_step'], det['step_wall_ms'], 'roofline frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])
print('cli', det['cli_end_to_end']['cold_entities_per_s'], det['cli_end_to_end']['warm_start_entities_per_s'], det['cli_subprocess']['cold_s'], det['cli_end_to_end_c5']['cold_s'], det['cli_end_to_end_ml20m_movie']['cold_s'])
print('fe', det['fixed_effect_eval']['ms_per_evaluation'], det['fixed_effect_eval']['frac_of_hbm_peak'], 'score', det['score_pass']['frac_of_hbm_peak'], 'handover', det['host_handover']['entities_per_s'])
PY
bash tools/r04_trace.sh > gpurun_out/trace/out.txt 2>&1; grep -E "^step 5" gpurun_out/trace/out.txt
