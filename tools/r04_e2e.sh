#!/bin/bash
# CLI end to end in process (tools/e2e_bench.py): three runs, then one with the reader's phase times (GDMIX_IO_TIMING) and the pipeline's
# timeline (E2E_TIMELINE: start / end / thread of every phase — who waits for whom)
mkdir -p gpurun_out/tt
cd /root/repo
for i in 1 2 3; do
  PYTHONPATH=. python tools/e2e_bench.py 1000000 8 2>&1 | grep -E "^cold start:|^warm start"
done
GDMIX_IO_TIMING=1 E2E_TIMELINE=1 PYTHONPATH=. python tools/e2e_bench.py 1000000 8 2>&1 | grep -v "amdgpu.ids\|^INFO" > gpurun_out/tt/e2e_timeline.txt
python - <<'PY'
import re, collections
txt = open('gpurun_out/tt/e2e_timeline.txt').read()
c = collections.defaultdict(list)
for m in re.finditer(r'\[gdmix_io\] (\S+)\s+([\d.]+) ms', txt):
    c[m.group(1)].append(float(m.group(2)))
print('reader phases (count, mean ms, max ms):', {k: (len(v), round(sum(v) / len(v), 1), round(max(v), 1)) for k, v in c.items()})
PY
grep -E "^cold start:|^warm" gpurun_out/tt/e2e_timeline.txt
awk '/^cold start: /{f=1} /^warm start/{f=0} f' gpurun_out/tt/e2e_timeline.txt | grep -E "\->" | cut -c1-120 | head -70
