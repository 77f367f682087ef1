#!/bin/bash
# A/B of builds of libgdmix_re.so (gdmix_amd/lib_<name>.so) on the fixed-effect bench: tools/fe_ab.sh name [name ...]
# DISTS="uniform zipf" (default both); other environment (GDMIX_FE_CHUNK ...) is passed through
export PYTHONPATH=. FE_BENCH_PATHS=stepping GDMIX_ALLOW_STALE_LIB=1
for v in "$@"; do cp gdmix_amd/lib_$v.so gdmix_amd/libgdmix_re.so; for d in ${DISTS:-uniform zipf}; do timeout 300 python tools/fe_bench.py ${SHAPE:-4000000 32 100000} $d 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v $d $GDMIX_FE_CHUNK', 'eval %.3f ms  rows %.3f cols %.3f  frac %.3f' % (d['ms_per_evaluation'], d['rows_pass_ms'], d['cols_pass_ms'], d['frac']))"; done; done
