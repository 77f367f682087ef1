#!/bin/bash
# A/B of the reader's 32-bit hand-over (gdmix_io_narrow, GDMIX_IO_WIRE=1 default) against the 64-bit arrays: GPU tests of the model path,
# then the CLI end to end on 1 M C2 entities in 8 partitions and on a C5-shaped directory, three runs each way.
mkdir -p gpurun_out/wire
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_rebalance.py -m gpu -q -x > gpurun_out/wire/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/wire/tests.log | cut -c1-300
for rep in 1 2 3; do
  for w in 1 0; do
    GDMIX_IO_WIRE=$w PYTHONPATH=. timeout 600 python tools/e2e_bench.py 1000000 8 c2 > gpurun_out/wire/c2_w${w}_$rep.log 2>&1
    echo "c2 wire=$w rep=$rep: $(grep -E "entities/s" gpurun_out/wire/c2_w${w}_$rep.log | tail -2 | cut -c1-60 | tr '\n' ' ' | cut -c1-400)"
  done
done
for w in 1 0; do
  GDMIX_IO_WIRE=$w PYTHONPATH=. timeout 600 python tools/e2e_bench.py 200000 8 zipf > gpurun_out/wire/zipf_w${w}.log 2>&1
  echo "zipf wire=$w: $(grep -E "entities/s" gpurun_out/wire/zipf_w${w}.log | tail -2 | cut -c1-60 | tr '\n' ' ' | cut -c1-400)"
done
