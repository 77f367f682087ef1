"""gdmix_amd — MI355X-native random-effect trainer behind gdmix-trainer's RandomEffect model API.

Only the random-effect hot path of linkedin/gdmix is implemented: hand-written HIP kernels
for gfx950 behind the C ABI in include/gdmix_re.h, and the host-side Python mirror of the reference's
RandomEffectLRLBFGSModel / REParams / driver / CLI for that path.
"""
__version__ = "0.1.0"

import os as _os

# A context deals its size classes over four streams (gdmix_re_set_spread) and a host pipeline keeps three contexts busy (model.py,
# bench.py's hand-over leg): twelve streams on the four hardware queues a process gets by default alias onto each other. Eight queues:
# hand-over leg 72 -> 79 M entities/s, nothing else moves (tools/r04_hwq.sh). Read by the HIP runtime when it initialises, i.e. at the
# first device call of the process — importing this package before that is enough; a value set by the caller wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
