#!/bin/bash
# where should the multi-workgroup team tiers start? (one-workgroup kernel with LDS vectors vs 2-CU teams)
for t in 16384 32768 65536 131072 262144; do
  for w in zipf c5share; do
    python bench.py --steps 2 --warmup 1 --workload $w --team-nnz $t --no-cpu-baseline --no-e2e --no-fe --no-cli 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w team_nnz=$t step %.1f ms solve %.1f' % (d['ms_per_step'], d['detail']['solve_ms_per_step']), [(n.split('kernel')[1].strip(),c,round(ms,1)) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c and 'team' in n])"
  done
done
