"""The shipped binaries carry the hash of their sources (missing 2): `gdmix_re_build_id()` / `gdmix_io_build_id()`.
`build.needs_build()` compares that hash — not modification times — and the loaders refuse a library that is not the build of the
sources it travels with. CPU only: hipcc cross-compiles, nothing is launched."""
import os
import shutil

import pytest

from gdmix_amd import build, solver
from gdmix_amd.io import native_reader


@pytest.fixture(scope="module", autouse=True)
def built():
    build.build_library()
    build.build_io_library()


def test_the_stamp_is_the_hash_of_the_sources_and_the_export_returns_it():
    assert build.embedded_id(build.LIB) == build.source_id()
    assert build.embedded_id(build.LIB, build.FLAGS_MARKER) == build.flags_id()
    assert solver.load_library().gdmix_re_build_id().decode() == build.source_id()
    assert build.embedded_id(build.IO_LIB) == build.io_source_id()
    assert native_reader.load_library().gdmix_io_build_id().decode() == build.io_source_id()
    assert not build.needs_build() and not build.needs_io_build()


def _tree_copy(tmp_path, monkeypatch):
    """A private copy of csrc/, include/ and the object cache, with build.py pointed at it."""
    pkg = tmp_path / "gdmix_amd"
    shutil.copytree(build.CSRC, pkg / "csrc")
    shutil.copytree(os.path.join(os.path.dirname(build.HERE), "include"), tmp_path / "include")
    if os.path.isdir(os.path.join(build.HERE, "build")):
        shutil.copytree(os.path.join(build.HERE, "build"), pkg / "build")
    shutil.copy2(build.LIB, pkg / "libgdmix_re.so")
    monkeypatch.setattr(build, "HERE", str(pkg))
    monkeypatch.setattr(build, "CSRC", str(pkg / "csrc"))
    monkeypatch.setattr(build, "LIB", str(pkg / "libgdmix_re.so"))
    return pkg


def test_an_edit_of_a_kernel_source_is_seen_whatever_the_mtimes_say_and_rebuilds(tmp_path, monkeypatch):
    pkg = _tree_copy(tmp_path, monkeypatch)
    assert not build.needs_build()
    src = pkg / "csrc" / "re_wire.hip"
    old_times = (os.path.getatime(src), os.path.getmtime(src))
    src.write_text(src.read_text() + "\n// an edit\n")
    os.utime(src, old_times)                       # an older-looking source: the round-5 mtime rule would have skipped it
    os.utime(pkg / "libgdmix_re.so", None)
    assert build.needs_build()
    # a header changes every unit's object key
    hdr = pkg / "csrc" / "re_device.hpp"
    key0 = build._object_key(str(src), build.FLAGS, [str(pkg / "csrc" / h) for h in build._headers()])
    hdr.write_text(hdr.read_text() + "\n// an edit\n")
    assert build._object_key(str(src), build.FLAGS, [str(pkg / "csrc" / h) for h in build._headers()]) != key0
    hdr.write_text(hdr.read_text()[:-len("\n// an edit\n")])
    # the rebuild compiles the one unit whose key changed (the others come from the copied cache) and stamps the new hash
    before = build.embedded_id(build.LIB)
    build.build_library()
    assert build.embedded_id(build.LIB) == build.source_id() != before
    assert not build.needs_build()


def test_other_flags_mean_another_build_but_the_same_sources(tmp_path, monkeypatch):
    _tree_copy(tmp_path, monkeypatch)
    monkeypatch.setenv("GDMIX_EXTRA_FLAGS", "-DGDMIX_SOME_EXPERIMENT=1")
    assert build.needs_build()                                              # for the builder: not the configured build
    assert build.check_library(build.LIB, build.source_id(), "sources") is None   # for the loader: still these sources


def _stale_copy(tmp_path, path):
    blob = open(path, "rb").read()
    i = blob.find(build.ID_MARKER) + len(build.ID_MARKER)
    out = tmp_path / os.path.basename(path)
    out.write_bytes(blob[:i] + b"0123456789abcdef" + blob[i + 16:])
    return str(out)


def test_a_stale_library_under_the_right_name_is_refused(tmp_path, monkeypatch):
    stale = _stale_copy(tmp_path, build.LIB)
    os.utime(stale, None)                                                   # newer than every source
    monkeypatch.setattr(solver, "LIB_PATH", stale)
    monkeypatch.setattr(solver, "_lib", None)
    monkeypatch.delenv("GDMIX_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(solver.GdmixReError, match="built from other sources"):
        solver.load_library()
    monkeypatch.setenv("GDMIX_ALLOW_STALE_LIB", "1")                        # what tools/ab.py sets to swap builds in
    assert solver.load_library().gdmix_re_build_id() == b"0123456789abcdef"
    monkeypatch.setattr(solver, "_lib", None)


def test_a_stale_io_library_is_refused(tmp_path, monkeypatch):
    stale = _stale_copy(tmp_path, build.IO_LIB)
    monkeypatch.setattr(native_reader, "LIB_PATH", stale)
    monkeypatch.setattr(native_reader, "_lib", None)
    monkeypatch.delenv("GDMIX_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(native_reader.GdmixIoError, match="built from other sources"):
        native_reader.load_library()
    monkeypatch.setattr(native_reader, "_lib", None)


def test_a_library_without_a_stamp_needs_a_build(tmp_path):
    p = tmp_path / "lib.so"
    p.write_bytes(b"\x7fELF" + b"\0" * 64)
    assert build.embedded_id(str(p)) is None
    assert build.check_library(str(p), build.source_id(), "sources") is not None
