#!/bin/bash
# more hardware queues per process (GPU_MAX_HW_QUEUES, ROCm's default 4): hand-over leg and C2 headline; "pkg" = left to gdmix_amd/__init__.py (8)
mkdir -p gpurun_out/tt
cd /root/repo
for q in 4 pkg 16 4 pkg 16; do
  if [ $q = pkg ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-fe --no-cli --no-other-workloads > gpurun_out/tt/hwq_$q.json 2> gpurun_out/tt/hwq_$q.err
  python - <<PY
import json
d = json.loads([l for l in open('gpurun_out/tt/hwq_$q.json') if l.startswith('{')][0])
det = d['detail']
print('hw queues $q: C2 ms/step', round(d['ms_per_step'], 3), 'handover', round(det['host_handover']['entities_per_s'] / 1e6, 1), 'M/s with index', round(det['host_handover'].get('with_feature_index', {}).get('entities_per_s', 0) / 1e6, 1) if isinstance(det['host_handover'].get('with_feature_index'), dict) else '', 'serial', round(det['host_handover']['serial_one_stream']['ms'], 2))
PY
done
