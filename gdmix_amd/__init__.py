"""gdmix_amd — MI355X-native random-effect trainer behind gdmix-trainer's RandomEffect model API.

Only the random-effect hot path of linkedin/gdmix is implemented: hand-written HIP kernels
for gfx950 behind the C ABI in include/gdmix_re.h, and the host-side Python mirror of the reference's
RandomEffectLRLBFGSModel / REParams / driver / CLI for that path.
"""
__version__ = "0.1.0"
