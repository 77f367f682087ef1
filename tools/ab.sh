#!/bin/bash
# A/B build variants on the GPU box: each argument is a quoted set of extra hipcc flags
for flags in "$@"; do
  echo "=== flags: $flags"
  GDMIX_EXTRA_FLAGS="$flags" python -m gdmix_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f ent/s  step %.2f ms  pack %.2f  solve %.2f  kernels %.2f' % (d['value'], d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['solve_kernel_ms_per_step']))
print('kernel ms per class:', [x for x in d['detail'].get('class_ms') if x])
"
done
