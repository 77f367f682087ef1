#!/bin/bash
# zipf workload at a larger scale: where does the time go per class
python bench.py --workload zipf --entities ${1:-1000000} --steps 2 --warmup 1 --no-cpu-baseline --no-e2e ${2} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f ent/s  step %.2f ms  pack %.2f  solve %.2f  kernels %.2f  N=%d Z=%d' % (d['value'], d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['solve_kernel_ms_per_step'], d['detail']['N'], d['detail']['Z']))
for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']):
    if c: print('  %-48s %8d  %9.3f ms  %9.1f ns/entity' % (n,c,ms,1e6*ms/c))
"
