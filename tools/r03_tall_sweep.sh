#!/bin/bash
mkdir -p gpurun_out/r03e
cd /root/repo
for mn in 16 32 48 64; do
  timeout 600 python bench.py --workload ml20m_user --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --tall-min-n $mn > gpurun_out/r03e/user_min$mn.json 2>/dev/null
  timeout 600 python bench.py --workload ml20m_movie --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --tall-min-n $mn > gpurun_out/r03e/movie_min$mn.json 2>/dev/null
done
