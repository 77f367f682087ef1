#!/bin/bash
# What could ANY restructuring of the L-BFGS direction gain on C2? Upper bound: the same kernels (M_REG = 10 pairs in registers, same
# occupancy) with the history capped at m pairs by the solver option: the two-loop recursion then runs 2m dependent reductions instead
# of 2 x min(k, 10). Time per evaluation = class ms / (mean evaluations of the class's entities).
for m in 10 5 2 1; do
  python bench.py --steps 5 --warmup 2 --lbfgs-m $m --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --project-ranks 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('m=$m step %.2f ms solve %.2f mean_nit %.2f mean_nfev %.2f' % (d['ms_per_step'], d['detail']['solve_ms_per_step'], d['detail']['mean_nit'], d['detail']['mean_nfev']))
for c in d['detail']['per_class']:
    if c['ms'] > 0.5: print('   %-45s ent %7d ms %6.3f nfev %5.2f  ns/entity/eval %6.2f' % (c['kernel'], c['entities'], c['ms'], c['mean_nfev'], c['ms']*1e6/c['entities']/c['mean_nfev']))"
done
