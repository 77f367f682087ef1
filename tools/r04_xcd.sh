#!/bin/bash
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team_tiers or device_wide or large_and_giant" > gpurun_out/r04e/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04e/tests.log
for x in 0 1 0 1; do
  for w in zipf c5share; do
    GDMIX_RE_XCD_BARRIER=$x python bench.py --steps 2 --warmup 1 --workload $w --no-cpu-baseline --no-e2e --no-fe --no-cli 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w xcd_barrier=$x step %.1f ms solve %.1f' % (d['ms_per_step'], d['detail']['solve_ms_per_step']), [(n.split('kernel')[1].strip(),c,round(ms,1)) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c and 'team' in n])"
  done
done
