#!/bin/bash
for flags in "$@"; do
  echo "=== flags: $flags"
  GDMIX_EXTRA_FLAGS="$flags" python -m gdmix_amd.build --force > /dev/null 2>&1 || { echo build failed; continue; }
  tools/zipf_scale.sh 1000000 | (head -1; tail -4)
done
