#!/bin/bash
export GDMIX_BENCH_LINE=full   # these scripts read the full result from stdout (bench.py prints the short line otherwise)
# Zipf-shaped partition: pack + solve per class (exploration workload of bench.py)
python bench.py --steps 3 --warmup 1 --workload zipf --no-cpu-baseline --no-e2e --no-fe --no-cli "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('zipf', d['value'], 'ent/s  step %.1f ms  pack %.1f  solve %.1f' % (d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'])); print([(n,c,ms) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c and ms > 1.0])"
