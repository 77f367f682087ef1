#!/bin/bash
# re-run single seeds of tools/fuzz_parity.py: bash tools/fuzz_seeds.sh 31 46 103
for s in "$@"; do PYTHONPATH=.:tests python tools/fuzz_parity.py 1 $s 2>&1 | grep -v amdgpu.ids | head -6; done
