#!/bin/bash
# CLI end to end in process: threads per native call (GDMIX_IO_THREADS, default 32) x partitions decoded ahead (GDMIX_PREFETCH_PARTITIONS, default 2)
cd /root/repo
for th in ${THREADS:-2 4 6 8 12}; do
  for pf in ${PREFETCH:-2 3}; do
    echo "io threads $th prefetch $pf: $(GDMIX_IO_THREADS=$th GDMIX_PREFETCH_PARTITIONS=$pf PYTHONPATH=. python tools/e2e_bench.py 1000000 8 2>&1 | grep -E "^cold start:|^warm start" | sed 's/ end to end.*//' | tr '\n' '|')"
  done
done
