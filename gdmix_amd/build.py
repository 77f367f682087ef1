"""In-tree build of libgdmix_re.so (hand-written HIP for gfx950 only, linked against the HIP runtime) and of
libgdmix_io.so (host C++: native TFRecord reader).

    python -m gdmix_amd.build            # build what was not compiled from the sources next to it (stamped hash, not mtimes)
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


# ---- build identity -------------------------------------------------------------------------------------------------------------
# Both libraries carry the hash of the sources they were compiled from (`gdmix_re_build_id()` / `gdmix_io_build_id()`; the string
# also sits in the file behind ID_MARKER, so that it can be read without dlopen). A library travels prebuilt to the GPU box next to
# its sources: what decides "is this binary the one these sources make" is that hash, never a modification time.
ID_MARKER = b"@(#)gdmix-build-id:"
FLAGS_MARKER = b"@(#)gdmix-build-flags:"


def _hash_files(paths, extra=b""):
    import hashlib
    h = hashlib.sha256(extra)
    for f in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def source_id():
    """Hash of every file libgdmix_re.so is compiled from (the eight .hip units, csrc/*.hpp, the two C headers)."""
    return _hash_files([os.path.join(CSRC, f) for f in SOURCES + _headers()])


def io_source_id():
    return _hash_files([os.path.join(CSRC, f) for f in IO_SOURCES + IO_HEADERS], " ".join(IO_FLAGS).encode())


def flags_id(extra=None):
    """Hash of the compiler flags (the defaults + GDMIX_EXTRA_FLAGS): stamped next to the source id, compared by needs_build()
    only — an A/B build of the same sources with other flags is still a build of these sources for the loader."""
    import hashlib
    extra = os.environ.get("GDMIX_EXTRA_FLAGS", "").split() if extra is None else extra
    return hashlib.sha256(" ".join(FLAGS + extra).encode()).hexdigest()[:16]


def _headers():
    return [os.path.basename(h) for h in glob.glob(os.path.join(CSRC, "*.hpp"))] + [os.path.join("..", "..", "include", "gdmix_re.h"), os.path.join("..", "..", "include", "gdmix_fe.h")]


def embedded_id(path, marker=ID_MARKER):
    """The build id stamped into a library file (None for a file without one: a build older than the stamp, or not ours)."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(marker)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(marker):j].decode("ascii", "replace")


def needs_build():
    """True when libgdmix_re.so is missing or was not compiled from the sources (and flags) next to it."""
    if os.environ.get("GDMIX_ALLOW_STALE_LIB", "0") == "1" and os.path.exists(LIB):
        return False      # an A/B build was swapped in under the product's name (tools/ab.py, tools/gpu_session.sh): leave it there
    return embedded_id(LIB) != source_id() or embedded_id(LIB, FLAGS_MARKER) != flags_id()


def needs_io_build():
    return embedded_id(IO_LIB) != io_source_id()


def check_library(path, want, what):
    """Loader-side gate (solver.load_library, io.native_reader.load_library): refuse a binary whose stamped id is not the hash of
    the sources next to it. GDMIX_ALLOW_STALE_LIB=1 turns the refusal into nothing (tools/ab.py swaps builds of other sources in);
    an installation without csrc/ has nothing to compare with and passes."""
    if os.environ.get("GDMIX_ALLOW_STALE_LIB", "0") == "1" or not os.path.isdir(CSRC):
        return None
    if callable(want):      # (the hash of the sources, computed only when there is something to compare with)
        try:
            want = want()
        except OSError:     # part of the sources is missing (a binary-only installation that kept csrc/ but not include/): nothing to compare with
            return None
    have = embedded_id(path)
    if have != want:
        return (f"{path} was built from other sources (its id {have}, {what} next to it hash to {want}): "
                "run `python -m gdmix_amd.build` (GDMIX_ALLOW_STALE_LIB=1 loads it anyway)")
    return None


def _object_key(src, flags, headers):
    return _hash_files([src] + headers, " ".join(flags).encode())


def build_library(force=False, verbose=False, out=None):
    """hipcc, one object per source in parallel (build/ next to the sources keeps them: a change to one kernel file recompiles
    that file), then one link. Objects are keyed by the hash of source, headers and flags (`<obj>.key`); `out` names another
    library file (A/B builds with GDMIX_EXTRA_FLAGS)."""
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
    headers = [os.path.join(CSRC, h) for h in _headers()]
    sid, fid = source_id(), flags_id(extra)

    def compile_one(f):
        src, obj = os.path.join(CSRC, f), os.path.join(objdir, f + ".o")
        key = _object_key(src, flags, headers)
        if not force and os.path.exists(obj) and os.path.exists(obj + ".key") and open(obj + ".key").read() == key:
            return obj
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(obj + ".key", "w") as fh:
            fh.write(key)
        return obj

    def compile_id():
        obj = os.path.join(objdir, f"build_id_{sid}_{fid}.o")
        if not os.path.exists(obj):
            for stale in glob.glob(os.path.join(objdir, "build_id_*.o")):
                os.remove(stale)
            cmd = [HIPCC, "-O1", "-fPIC", "-fvisibility=hidden", "-x", "c++", "-c", os.path.join(CSRC, "build_id.inc"), "-o", obj,
                   "-DGDMIX_ID_FN=gdmix_re_build_id", f'-DGDMIX_BUILD_ID="{sid}"', f'-DGDMIX_BUILD_FLAGS="{fid}"']
            subprocess.check_call(cmd)
        return obj
    with ThreadPoolExecutor(len(SOURCES) + 1) as ex:
        idobj = ex.submit(compile_id)
        objs = list(ex.map(compile_one, SOURCES)) + [idobj.result()]
    tmp = out + ".tmp"
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    return out


def build_io_library(force=False, verbose=False):
    if not force and not needs_io_build():
        return IO_LIB
    sid = io_source_id()
    cmd = [CXX] + IO_FLAGS + [os.path.join(CSRC, f) for f in IO_SOURCES] + ["-x", "c++", os.path.join(CSRC, "build_id.inc"),
                                                                           "-DGDMIX_ID_FN=gdmix_io_build_id", f'-DGDMIX_BUILD_ID="{sid}"',
                                                                           '-DGDMIX_BUILD_FLAGS="io"', "-lz", "-o", IO_LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(IO_LIB + ".tmp", IO_LIB)
    return IO_LIB


if __name__ == "__main__":
    force = "--force" in sys.argv
    out = None
    if "--out" in sys.argv:          # python -m gdmix_amd.build --out gdmix_amd/lib_x.so   (with GDMIX_EXTRA_FLAGS: an A/B build)
        out = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
    if "--io-only" not in sys.argv:
        print(build_library(force=force, verbose=True, out=out))
    print(build_io_library(force=force, verbose=True))
