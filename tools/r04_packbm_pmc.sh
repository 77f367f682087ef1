#!/bin/bash
# VALU instructions and duration of pack_entity_kernel<256> with the bitmap path on / off (C2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-fe --no-cli --no-other-workloads --no-alone --project-ranks 0"
for t in 0 1; do
  O=gpurun_out/prof_bm$t; rm -rf $O; mkdir -p $O
  GDMIX_PACK_BITMAP=$t rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
  GDMIX_PACK_BITMAP=$t rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o q -- $CMD > $O/sq.log 2>&1
  python tools/prof_summary.py --stats $(ls $O/stats/*.db | head -1) --pmc $(ls $O/sq/*.db | head -1) 2>&1 | grep "pack_entity_kernel<256" | cut -c1-200 | sed "s/^/bitmap=$t /"
  find $O -name "*.db" -delete
done
