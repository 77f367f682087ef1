#!/bin/bash
# small tall class: wavefronts per workgroup (x workgroups per CU = 8 wavefronts)
mkdir -p gpurun_out/r03e
cd /root/repo
cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so
for nw in 1 2 4; do
  GDMIX_EXTRA_FLAGS="-DGDMIX_TALL_NW_SMALL=$nw" python -m gdmix_amd.build --force > /dev/null 2>&1 || exit 1
  for mn in 64 128; do
    timeout 600 python bench.py --workload ml20m_user --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --tall-min-n $mn --tall-split-n 4096 > gpurun_out/r03e/user_nw${nw}_min$mn.json 2>/dev/null
  done
  timeout 600 python bench.py --workload ml20m_movie --steps 3 --warmup 1 --no-e2e --no-cpu-baseline --tall-min-n 64 --tall-split-n 4096 > gpurun_out/r03e/movie_nw${nw}.json 2>/dev/null
done
cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so
