"""In-tree build of libgdmix_re.so (hand-written HIP for gfx950 only, linked against the HIP runtime) and of
libgdmix_io.so (host C++: native TFRecord reader).

    python -m gdmix_amd.build            # build what is older than its sources
    python -m gdmix_amd.build --force
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgdmix_re.so")
SOURCES = ["re_api.hip", "re_solve.hip", "re_solve_tall.hip", "re_pack.hip", "re_pack_big.hip", "re_wire.hip", "re_variance_big.hip", "fe_solve.hip"]
HEADERS = [os.path.basename(h) for h in glob.glob(os.path.join(CSRC, "*.hpp"))] + [os.path.join("..", "..", "include", "gdmix_re.h"), os.path.join("..", "..", "include", "gdmix_fe.h")]
IO_LIB = os.path.join(HERE, "libgdmix_io.so")
IO_SOURCES = ["io_reader.cpp", "io_avro.cpp"]
IO_HEADERS = [os.path.join("..", "..", "include", "gdmix_io.h")]
CXX = os.environ.get("CXX", "g++")
IO_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-pthread"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -disable-machine-licm: MachineLICM hoists the ~35 fp64 polynomial constants of exp/log out of the solver's
# main loop, where they stay live in VGPRs across the whole solve and cost a wave of occupancy.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-mllvm", "-disable-machine-licm"]


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources of the random-effect kernels (csrc/re_*): what PMC evidence collected on one
    build of the library is stamped with (profiles/latest_traffic.json), so that bench.py can tell a stale file from a current one."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, "re_*"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False, out=None):
    """hipcc, one object per source in parallel (build/ next to the sources keeps them: a change to one kernel file recompiles
    that file), then one link. Objects are keyed by source, headers and flags; `out` names another library file (A/B builds
    with GDMIX_EXTRA_FLAGS)."""
    out = out or LIB
    if not force and out == LIB and not needs_build():
        return LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    extra = os.environ.get("GDMIX_EXTRA_FLAGS", "").split()   # tuning experiments only
    flags = [f for f in FLAGS if f != "-shared"] + extra
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)

    def compile_one(f):
        src, obj = os.path.join(CSRC, f), os.path.join(objdir, f + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def build_io_library(force=False, verbose=False):
    if not force and os.path.exists(IO_LIB):
        t = os.path.getmtime(IO_LIB)
        if not any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in IO_SOURCES + IO_HEADERS):
            return IO_LIB
    cmd = [CXX] + IO_FLAGS + [os.path.join(CSRC, f) for f in IO_SOURCES] + ["-lz", "-o", IO_LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return IO_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    out = None
    if "--out" in sys.argv:          # python -m gdmix_amd.build --out gdmix_amd/lib_x.so   (with GDMIX_EXTRA_FLAGS: an A/B build)
        out = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
    if "--io-only" not in sys.argv:
        print(build_library(force=force, verbose=True, out=out))
    print(build_io_library(force=force, verbose=True))
