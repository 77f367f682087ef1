#!/bin/bash
mkdir -p gpurun_out/r04p
timeout 900 python -m pytest tests/test_fixed_effect.py tests/test_fe_model.py -m gpu -x -q > gpurun_out/r04p/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r04p/tests.log | cut -c1-300
for d in uniform zipf zipf; do PYTHONPATH=. FE_BENCH_PATHS=stepping python tools/fe_bench.py 4000000 32 100000 $d 2>/dev/null | tail -3 | cut -c1-400; done
echo "== counters zipf"; bash tools/fe_prof_args.sh gpurun_out/r04p/fe_zipf 4000000 32 100000 zipf | grep "fe_scatter\|fe_hot\|fe_finish" | head -8
grep -E "fe_scatter_kernel<false.*(SQ_LDS_BANK_CONFLICT|SQ_WAIT_INST_LDS|FETCH_SIZE|WRITE_SIZE|SQ_ACTIVE_INST_LDS)" gpurun_out/r04p/fe_zipf/summary.txt | cut -c1-150
