#!/bin/bash
for w in "c5mean 200000" "zipf 100000" "ml_user 20000" "ml_movie 27000"; do
  set -- $w
  echo "=== $1 ($2 entities)"
  python bench.py --workload $1 --entities $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f ent/s  step %.2f ms  pack %.2f  solve %.2f  kernels %.2f  N=%d Z=%d nit %.1f' % (d['value'], d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['solve_kernel_ms_per_step'], d['detail']['N'], d['detail']['Z'], d['detail']['mean_nit']))
print([(n,c,ms) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c])
"
done
