#!/bin/bash
# round 3, first GPU session: the new full-size tests, the bench harness, every bench workload
mkdir -p gpurun_out/r03a
cd /root/repo
python -c "import __graft_entry__ as g; g.build()" || exit 1
timeout 2400 python -m pytest tests/test_gpu_full_size.py tests/test_bench_harness.py -m gpu -q -s -x > gpurun_out/r03a/tests_full_size.log 2>&1
echo "full-size tests rc=$?" | tee -a gpurun_out/r03a/tests_full_size.log
tail -40 gpurun_out/r03a/tests_full_size.log
for w in ml20m_user ml20m_movie c5share zipf; do
  timeout 900 python bench.py --workload $w --steps 2 --warmup 1 --no-e2e > gpurun_out/r03a/bench_$w.json 2> gpurun_out/r03a/bench_$w.err
  echo "$w rc=$?"
done
timeout 900 python bench.py > gpurun_out/r03a/bench_c2.json 2> gpurun_out/r03a/bench_c2.err; echo "c2 rc=$?"
