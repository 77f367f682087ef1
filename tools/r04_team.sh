#!/bin/bash
# round 4: the LDS vectors of the one-workgroup team kernel — parity, then A/B on the Zipf workloads
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team_lds or block_kernel or team_tiers or device_wide or reproducible" > gpurun_out/r04b/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r04b/tests.log
for kb in 0 200; do
  echo "== GDMIX_TEAM_ARENA_KB=$kb"
  GDMIX_TEAM_ARENA_KB=$kb bash tools/zipf_quick.sh --no-other-workloads
  GDMIX_TEAM_ARENA_KB=$kb python bench.py --steps 2 --warmup 1 --workload c5share --no-cpu-baseline --no-e2e --no-fe --no-cli 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('c5share', d['value'], 'ent/s  step %.1f ms' % d['ms_per_step']); print([(n,c,ms) for (n,c),ms in zip(d['detail']['classes'], d['detail']['class_ms']) if c and ms > 5.0])"
done
