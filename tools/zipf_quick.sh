#!/bin/bash
run() {
  python bench.py --workload zipf --entities 100000 --steps 2 --warmup 1 --no-cpu-baseline $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f ent/s  step %.2f ms  pack %.2f  solve %.2f  kernels %.2f' % (d['value'], d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['solve_kernel_ms_per_step']))
print([(n,c,ms) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c and ms > 1.5])
"
}
for t in 8 16 32; do for tn in 16384 8192 4096; do
  echo "=== teams $t team_nnz $tn"
  GDMIX_RE_TEAMS=$t run "--team-nnz $tn"
done; done
