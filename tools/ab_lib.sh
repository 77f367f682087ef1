#!/bin/bash
# A/B of two builds of libgdmix_re.so on the C2 bench: tools/ab_lib.sh gdmix_amd/lib_a.so gdmix_amd/lib_b.so [bench args]
A=$1; B=$2; shift 2
for rep in 1 2; do for L in $A $B; do cp $L gdmix_amd/libgdmix_re.so; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fe --no-cli "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', 'step %.3f ms  solve %.3f' % (d['ms_per_step'], d['detail']['solve_ms_per_step']), [round(x,3) for x in d['detail']['class_ms'] if x>0.2])"; done; done
