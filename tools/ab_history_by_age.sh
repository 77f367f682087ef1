#!/bin/bash
# A/B of the group kernels' history split (GDMIX_QUAD_NOLD = pairs kept in LDS) on the C2 bench. Build the variants first (CPU box):
#   for n in 0 2 3 4; do GDMIX_EXTRA_FLAGS="-DGDMIX_QUAD_NOLD=$n" python -m gdmix_amd.build --force; cp gdmix_amd/libgdmix_re.so gdmix_amd/lib_nold$n.so; done
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
cp gdmix_amd/libgdmix_re.so /tmp/keep.so
for rep in 1 2; do for n in 0 2 3 4; do cp gdmix_amd/lib_nold$n.so gdmix_amd/libgdmix_re.so; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('nold $n', 'step %.3f ms  solve %.3f' % (d['ms_per_step'], d['detail']['solve_ms_per_step']), [round(x,3) for x in d['detail']['class_ms'] if x>0.2])"; done; done | tee gpurun_out/r03i/ab_nold.txt
cp /tmp/keep.so gdmix_amd/libgdmix_re.so
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
