cd /root/repo
mkdir -p gpurun_out/r04c
SIDES="3 1 3" bash tools/r04_handover.sh
for n in 3 2 1 0; do
GDMIX_RE_SIDE_STREAM=$n timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pack or tall or routing or fixture_l2 or ragged" > gpurun_out/r04c/tests_$n.log 2>&1
echo "side=$n tests rc=$?"; tail -1 gpurun_out/r04c/tests_$n.log | cut -c1-200
done
SKIP_TESTS=1 ADAPT_LIST="192" bash tools/r04_adapt.sh
