#!/bin/bash
python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f ent/s  step %.2f ms  pack %.2f  solve %.2f  kernels %.2f  nit %.2f nfev %.2f' % (d['value'], d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['solve_kernel_ms_per_step'], d['detail']['mean_nit'], d['detail']['mean_nfev']))
for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']):
    if c: print('  %-48s %8d  %7.3f ms  %6.1f ns/entity' % (n,c,ms,1e6*ms/c))
"
