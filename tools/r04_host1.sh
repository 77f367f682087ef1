#!/bin/bash
mkdir -p gpurun_out/r04g
python tools/cold_start_parts.py 2>&1 | tail -25
echo "--- again (page cache warm)"
python tools/cold_start_parts.py 2>&1 | tail -25
ls -la gdmix_amd/*.so
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for p in d['detail']['strong_projection']:
    print(p['workload'], 'ms', round(p['ms'],2), [(round(r['ms_per_step'],2), round(r['pack_ms'],2), round(r['solve_kernel_ms'],2)) for r in p['per_rank']][:4])"
