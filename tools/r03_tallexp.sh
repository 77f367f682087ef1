#!/bin/bash
# what bounds the tall kernel's pass: builds without the LDS adds / without the logistic terms (timing only, wrong results)
mkdir -p gpurun_out/r03d
cd /root/repo
cp gdmix_amd/libgdmix_re.so /tmp/lib_keep.so
for v in BASE NO_ADD NO_LOGISTIC; do
  GDMIX_EXTRA_FLAGS="-DGDMIX_TALL_PROFILE -DGDMIX_TALL_EXP_$v" python -m gdmix_amd.build --force > /dev/null 2>&1 || exit 1
  echo "== $v" >> gpurun_out/r03d/exp.txt
  timeout 600 python bench.py --workload ml20m_movie --steps 1 --warmup 0 --no-e2e --no-cpu-baseline 2>&1 | grep "^tall" | sort -t= -k2 -n | awk 'NR==1||NR==2||NR==4||NR==8||NR==12||NR==20' >> gpurun_out/r03d/exp.txt
done
cp /tmp/lib_keep.so gdmix_amd/libgdmix_re.so
