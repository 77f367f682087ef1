#!/bin/bash
# whole GPU suite + smoke + the default bench line (what the driver runs at the end of a round)
mkdir -p gpurun_out/r04y
timeout 3000 python -m pytest tests -m gpu -q -x > gpurun_out/r04y/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04y/tests.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r04y/bench.json 2> gpurun_out/r04y/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04y/bench.json') if l.startswith('{')][0])
det = d['detail']
print('value', d['value'], 'ms', d['ms_per_step'])
for p in det['strong_projection'] or []:
    print(p['workload'], 'ms', round(p['ms'], 2), 'imb', round(p['imbalance'], 3), (p.get('partition_rounds') or {}).get('mean_imbalance'), (p.get('partition_rounds') or {}).get('worst_imbalance'))
    if p.get('partition_rounds'):
        for r in p['partition_rounds']['rounds']: print('   ', r)
for k, v in (det['workloads'] or {}).items():
    print(k, {a: b for a, b in v.items() if a in ('ms_per_step', 'entities_per_s')}, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v['roofline'].items() if a in ('kernel', 'avg_launch_ms', 'achieved_GBps', 'frac_of_hbm_peak', 'restreamed_GBps', 'restreamed_frac_of_hbm_peak')})
print('cli', det['cli_end_to_end']['cold_entities_per_s'], det['cli_end_to_end']['warm_start_entities_per_s'], det['cli_subprocess']['cold_s'], det['cli_end_to_end_c5']['cold_s'])
print('fe', det['fixed_effect_eval']['ms_per_evaluation'], 'handover', det['host_handover']['entities_per_s'])
PY
