#!/bin/bash
# a lean tall class of fewer than R x (CUs x 12) entities runs inside the general one-wavefront launch: R = 4 (round 3), 2, 1, 0 (never)
for R in 4 2 1 0; do
  for w in ml20m_user ml20m_movie; do
  GDMIX_RE_LEAN_MERGE_ROUNDS=$R python bench.py --gpus 1 --scaling strong --workload $w --steps 5 --warmup 2 --no-rebalance 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('R=$R $w whole population: %.2f ms' % d['ms_per_step'])"
  done
  GDMIX_RE_LEAN_MERGE_ROUNDS=$R python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli --c5-entities 50000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for p in d['detail']['strong_projection'][:2]:
    print('R=$R', p['workload'], '8 shares: ms', round(p['ms'],2), [round(r['ms_per_step'],2) for r in p['per_rank']], p['per_rank'][0]['largest_launches'][:3])"
done
