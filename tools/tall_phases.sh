#!/bin/bash
# per-phase times of the tall kernel (build with -DGDMIX_TALL_PROFILE into a side library)
mkdir -p gpurun_out/r03d
cd /root/repo
cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so
GDMIX_EXTRA_FLAGS="-DGDMIX_TALL_PROFILE $TALL_EXTRA" python -m gdmix_amd.build --force > /dev/null 2>&1 || exit 1
for w in ${TALL_WORKLOADS:-ml20m_user ml20m_movie}; do
  timeout 600 python bench.py --workload $w --steps 1 --warmup 0 --no-e2e --no-cpu-baseline 2>&1 | grep "^tall" | sort -t= -k2 -n | awk 'NR<=3||NR%3==0' | head -24 > gpurun_out/r03d/prof_$w.txt
done
cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so
