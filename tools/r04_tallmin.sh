#!/bin/bash
# where should the tall kernel start? (entities with p <= 64: group kernels up to their sample caps vs the tall kernel from n >= tall_min_n)
for w in ml20m_user ml20m_movie; do
for t in 32 65 129 257 513; do
  python bench.py --steps 3 --warmup 1 --workload $w --tall-min-n $t --no-cpu-baseline --no-e2e --no-fe --no-cli 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w tall_min_n=$t step %.2f ms pack %.2f solve %.2f' % (d['ms_per_step'], d['detail']['pack_ms_per_step'], d['detail']['solve_ms_per_step']), [(n.replace('re_solve_','').replace('_kernel',''),c,round(ms,2)) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c and ms > 0.3])"
done; done
