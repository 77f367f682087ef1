#!/bin/bash
# pack A/B on the GPU box: parity tests of the pack first, then bench with the old and the new library, then the kernels' profile
set -u
cd "$(dirname "$0")/.."
export PYTHONPATH=.:tests
out=gpurun_out/${1:-pack}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt
timeout 1500 python tools/ab.py --lib gdmix_amd/lib_pack_a.so,gdmix_amd/lib_pack_b.so --workloads ${2:-c2} --reps 2 --out $out/ab > $out/ab.txt 2>&1
tail -8 $out/ab.txt
cp gdmix_amd/lib_pack_b.so gdmix_amd/libgdmix_re.so
bash tools/pack_prof.sh ${1:-pack}/prof c2 > $out/prof.txt 2>&1
head -30 $out/prof/summary.txt | grep -i "pack\|scan"
