#!/bin/bash
# host hand-over regression (76 M ent/s in BENCH_r02, 58 M in BENCH_r03): the side stream of round 3?
for v in ${SIDES:-3 1 3 1 0}; do
  GDMIX_RE_SIDE_STREAM=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fe --no-cli --no-other-workloads --no-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['detail']['host_handover']
print('side_stream=$v  handover %.1f M ent/s (%.2f ms/partition)  with index %.1f M  serial %.2f ms   step %.2f ms' % (h['entities_per_s']/1e6, h['ms_per_partition'], h['with_feature_index']['entities_per_s']/1e6, h['serial_one_stream']['ms'], d['ms_per_step']))"
done
