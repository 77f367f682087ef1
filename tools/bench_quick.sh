#!/bin/bash
# quick GPU check: parity tests + one bench line (summary only)
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f ent/s  step %.2f ms  pack %.2f  solve %.2f  kernels %.2f' % (d['value'], d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['solve_kernel_ms_per_step']))
print([(n,c) for n,c in d['detail']['classes'] if c])
print('kernel ms per class:', d['detail'].get('class_ms'))
"
