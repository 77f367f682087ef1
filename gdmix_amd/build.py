"""In-tree build of libgdmix_re.so: hand-written HIP for gfx950 only, linked against the HIP runtime.

    python -m gdmix_amd.build            # build if sources are newer than the library
    python -m gdmix_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgdmix_re.so")
SOURCES = ["re_api.hip", "re_solve.hip", "re_pack.hip", "re_pack_big.hip"]
HEADERS = ["re_device.hpp", "re_solve_core.hpp", "re_internal.hpp", os.path.join("..", "..", "include", "gdmix_re.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -disable-machine-licm: MachineLICM hoists the ~35 fp64 polynomial constants of exp/log out of the solver's
# main loop, where they stay live in VGPRs across the whole solve and cost a wave of occupancy.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-mllvm", "-disable-machine-licm"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    extra = os.environ.get("GDMIX_EXTRA_FLAGS", "").split()   # tuning experiments only
    cmd = [HIPCC] + FLAGS + extra + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
