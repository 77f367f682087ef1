#!/bin/bash
mkdir -p gpurun_out/r04c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
cp gdmix_amd/libgdmix_re.so /tmp/libgdmix_re.keep
GDMIX_EXTRA_FLAGS="-DGDMIX_TEAM_PROFILE" python -m gdmix_amd.build --force > gpurun_out/r04c/build.log 2>&1
python bench.py --steps 1 --warmup 1 --workload zipf --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads > gpurun_out/r04c/zipf_prof.out 2>/dev/null
grep "^team" gpurun_out/r04c/zipf_prof.out | sort -t= -k3 -n | awk 'NR%4==1' | tail -60
cp /tmp/libgdmix_re.keep gdmix_amd/libgdmix_re.so
